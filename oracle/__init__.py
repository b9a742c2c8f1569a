"""CPU oracle for the AV-VAD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (torch fp32 functional ops / explicit Python
loops) of the algorithm of the reference's per-frame classification path
(sp-uhh/audio-visual-vad, ``packages/models/*``, ``packages/utils.py``,
``packages/processing/stft.py``).  Every function cites the reference file:line
it restates.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and only as the *checker* / reported
baseline.  The product path (``audio-visual-vad_amd/``) never imports it and
raises when the HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * WaveNet encoder, LSTM heads, losses, collates, count sketch: pinned against
    outputs of the reference itself, generated in the build container by
    ``tools/gen_golden.py`` (imports ``/root/reference``) and committed under
    ``tests/golden/``.
  * ResNet-18 trunk arithmetic: the reference delegates it to torchvision
    (un-vendored, un-pinned, not installed here) -> "parity unpinned" by the
    reference; the restatement in ``oracle/resnet18.py`` follows torchvision's
    published ResNet-18 structure and is pinned only by parameter count, key
    names, shape chain and the reference's own class bodies executing on top
    of it.
  * STFT front-end / compact bilinear pooling forward: the reference code
    calls APIs removed from torch 2.x (``torch.rfft``, legacy ``torch.stft``),
    so it cannot run here; restated from the text and pinned against
    independent definitions (naive DFT / naive outer-product sketch).
"""
