"""Training / evaluation loops shared by the ``scripts/train_*.py`` and ``scripts/evaluate_*.py`` entry points.

They mirror the loop bodies of the reference (``scripts/train_AV_net.py:252-448``, ``train_audio_net.py:190-372``,
``train_video_net.py:182-319``, ``evaluate_AV_net.py:148-250``): per batch -- move to the GPU, standardise with the
train-set statistics when given (``std_norm``: ``(x - mean.T) / (std + eps).T``, ``train_AV_net.py:286-291``), forward,
per-sequence masked BCE summed over the batch, backward, Adam, per-sequence accuracy/precision/recall/F1 -- with the
reference's Python loops over the batch replaced by single fused calls, ``nn.DataParallel`` replaced by one process per
GPU + bucketed RCCL all-reduce, and a synthetic data source for training (the reference's HDF5 readers are out of scope,
SURVEY.md 2.1; h5py and torchaudio are not installed in this image).  The per-utterance evaluator of the audio network
(``process_utt``, ``evaluate_audio_net.py:107-180``) runs the reference's whole chain on real waveforms: peak
normalisation -> STFT -> power -> log -> crop to the label length -> standardise -> classifier -> sigmoid -> threshold."""
import os
import time

import torch

from . import dist as avd
from . import ops
from .optim import FlatAdam

EPS = 1e-8


class Stats:
    """Train-set mean / std used by ``std_norm`` (the reference reads them from HDF5 and saves ``trainset_*_mean.npy``,
    ``train_AV_net.py:206-231``): audio (513,1) per-bin vectors, video (1,1) scalars."""

    def __init__(self, audio_mean=None, audio_std=None, video_mean=None, video_std=None, eps=EPS):
        self.eps = eps
        self._raw = dict(audio_mean=audio_mean, audio_std=audio_std, video_mean=video_mean, video_std=video_std)
        self._dev = {}

    @classmethod
    def load(cls, model_dir, eps=EPS):
        """``trainset_{audio,video}_{mean,std}.npy`` as the reference's training script writes them."""
        import numpy as np
        kw = {}
        for k in ("audio_mean", "audio_std", "video_mean", "video_std"):
            path = os.path.join(model_dir, "trainset_%s.npy" % k)
            if os.path.exists(path):
                kw[k] = np.load(path, allow_pickle=False)
        return cls(eps=eps, **kw)

    def get(self, key, device):
        if self._raw.get(key) is None:
            return None
        if (key, device) not in self._dev:
            self._dev[(key, device)] = torch.as_tensor(self._raw[key], dtype=torch.float32).reshape(-1).to(device)
        return self._dev[(key, device)]

    def audio(self, x):
        m, s = self.get("audio_mean", x.device), self.get("audio_std", x.device)
        return x if m is None else ops.standardize(x, m, s, self.eps)

    def video(self, v):
        m, s = self.get("video_mean", v.device), self.get("video_std", v.device)
        return v if m is None else ops.standardize(v, m, s, self.eps)


class SyntheticAV(torch.utils.data.Dataset):
    """Items shaped like the reference's datasets return them (``data_handling.py:387-495``): audio features
    (513, T) or a waveform (L,), video (67, 67, T), target (1, T), [L,] T -- ragged T per item."""

    def __init__(self, n_items, kind, t_min=8, t_max=16, waveform=False, rf=2048, seed=0):
        self.n, self.kind, self.t_min, self.t_max, self.waveform, self.rf, self.seed = n_items, kind, t_min, t_max, waveform, rf, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        T = int(torch.randint(self.t_min, self.t_max + 1, (1,), generator=g))
        y = (torch.rand(1, T, generator=g) > 0.5).float()
        item = []
        L = T * 256 + self.rf - 1
        if self.kind in ("audio", "av"):
            if self.waveform:
                w = torch.rand(L, generator=g) * 2 - 1
                item.append(w / w.abs().max())
            else:
                item.append(torch.randn(513, T, generator=g))
        if self.kind in ("video", "av"):
            item.append(torch.randn(67, 67, T, generator=g))
        item.append(y)
        if self.waveform and self.kind != "video":
            item.append(L)
        item.append(T)
        return tuple(item)


def pick_collate(kind, waveform):
    from packages import utils as U
    if kind == "audio":
        return U.collate_many2many_audio_waveform if waveform else U.collate_many2many_audio
    if kind == "video":
        return U.collate_many2many_video
    return U.collate_many2many_AV_waveform if waveform else U.collate_many2many_AV


def forward_batch(model, kind, batch, device, waveform, stats=None):
    """H2D, ``std_norm`` standardisation (spectrogram features and video; raw waveforms are not standardised in the
    reference either), forward."""
    lengths = batch[0].to(device)
    data = [t.to(device, non_blocking=True) for t in batch[1:]]
    y = data[-1]
    if kind == "audio":
        x = data[0].unsqueeze(1) if waveform else (stats.audio(data[0]) if stats else data[0])
        return lengths, model(x, lengths), y
    if kind == "video":
        return lengths, model(stats.video(data[0]) if stats else data[0], lengths), y
    a = data[0].unsqueeze(1) if waveform else (stats.audio(data[0]) if stats else data[0])
    return lengths, model(a, stats.video(data[1]) if stats else data[1], lengths), y


def run_epoch(model, kind, loader, device, waveform, opt=None, reducer=None, log=None, log_interval=10, stats=None):
    from packages.models.utils import batch_binary_cross_entropy, batch_f1
    train = opt is not None
    model.train(train)
    tot = dict(loss=0.0, acc=0.0, prec=0.0, rec=0.0, f1=0.0, n=0)
    for i, batch in enumerate(loader):
        lengths, logits, y = forward_batch(model, kind, batch, device, waveform, stats)
        loss = batch_binary_cross_entropy(logits, y, lengths, EPS)       # sum over sequences (train_AV_net.py:298-302)
        if train:
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            opt.zero_grad()
        hard = (torch.sigmoid(logits.detach()) > 0.5).int()
        acc, prec, rec, f1 = batch_f1(hard, y.long(), lengths, EPS)
        for k, v in zip(("loss", "acc", "prec", "rec", "f1"), (loss.detach(), acc, prec, rec, f1)):
            tot[k] += float(v)
        tot["n"] += 1
        if log and i % log_interval == 0:
            log("%s batch %4d  loss %.3f  acc %.3f  prec %.3f  rec %.3f  f1 %.3f"
                % ("train" if train else "valid", i, float(loss.detach()), float(acc), float(prec), float(rec), float(f1)))
    n = max(tot.pop("n"), 1)
    return {k: v / n for k, v in tot.items()}


def train_main(kind, make_model, model_name, waveform=False, epochs=1, batch_size=16, n_items=64, lr=1e-4,
               freeze_features=False, out_dir=None, stats=None):
    """The body of ``scripts/train_{audio,video,AV}_net.py``; settings come from the caller's module-level constants
    (the reference's "config system") and may be overridden by AVVAD_* environment variables."""
    epochs = int(os.environ.get("AVVAD_EPOCHS", epochs))
    n_items = int(os.environ.get("AVVAD_ITEMS", n_items))
    batch_size = int(os.environ.get("AVVAD_BATCH", batch_size))
    rank, world, local = avd.init_from_env("nccl")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    model = make_model().to(device)
    if freeze_features:                      # train_AV_net.py:241-245
        for name, child in model.named_children():
            if name == "features":
                for p in child.parameters():
                    p.requires_grad = False
    opt = FlatAdam(model.parameters(), lr=lr, betas=(0.9, 0.999))
    reducer = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets,
                                names=[n for n, q in model.named_parameters() if q.requires_grad]) if world > 1 else None
    collate = pick_collate(kind, waveform)
    per_rank = n_items // world
    ds_train = SyntheticAV(per_rank, kind, waveform=waveform, seed=1 + rank)
    ds_valid = SyntheticAV(max(per_rank // 4, batch_size), kind, waveform=waveform, seed=1000 + rank)
    mk = lambda ds, sh: torch.utils.data.DataLoader(ds, batch_size=batch_size // world or 1, shuffle=sh, collate_fn=collate)
    out_dir = out_dir or os.path.join("models", model_name)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    logf = open(os.path.join(out_dir, "output_batch.log"), "a") if rank == 0 else None

    def log(msg):
        if rank == 0:
            print(msg, flush=True)
            print(msg, file=logf, flush=True)

    from packages.utils import count_parameters
    log("- Number of learnable parameters: {}".format(count_parameters(model)))
    for epoch in range(1, epochs + 1):
        t0 = time.perf_counter()
        tr = run_epoch(model, kind, mk(ds_train, True), device, waveform, opt, reducer, log, stats=stats)
        with torch.no_grad():
            va = run_epoch(model, kind, mk(ds_valid, False), device, waveform, stats=stats)
        log("====> Epoch: {:2d}  train loss {:.3f} f1 {:.3f} | valid loss {:.3f} f1 {:.3f} | {:.1f} s".format(
            epoch, tr["loss"], tr["f1"], va["loss"], va["f1"], time.perf_counter() - t0))
        if rank == 0:                       # same checkpoint naming as train_AV_net.py:443-448
            torch.save(model.state_dict(), os.path.join(out_dir, "Video_Net_epoch_{:03d}_vloss_{:.2f}.pt".format(epoch, va["loss"])))
    return model


def load_waveform(path):
    """16 kHz mono utterance as a float32 tensor: ``.wav`` (int16 PCM scaled by 1/32768 like ``torchaudio.load``) or an
    ``.npz`` holding the int16 ``samples`` of one (the committed test fixture)."""
    import numpy as np
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        x, fs = z["samples"], int(z["fs"])
    else:
        from scipy.io import wavfile
        fs, x = wavfile.read(path)
    if x.ndim > 1:
        x = x[:, 0]                      # 1 channel (evaluate_audio_net.py:120)
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), fs


def audio_features(x_t, stats=None, n_label_frames=None, fs=16e3, wlen_sec=64e-3, hop_percent=0.25, eps=EPS, std_norm=True):
    """``process_utt`` up to the classifier input (``evaluate_audio_net.py:122-163``) on the GPU: x / max|x| -> STFT
    (Hann 1024 / hop 256, center=False, one-hop end pad) -> re^2 + im^2 -> log(. + eps) -> crop to the label length ->
    (x - mean.T) / (std + eps).T.  Peak normalisation is its own kernel; everything behind the DFT is one epilogue pass.
    x_t (L,) on the GPU -> (1, T, 513)."""
    nfft = int(wlen_sec * fs)
    x_t = ops.peak_normalize(x_t)
    mean = std = None
    if std_norm and stats is not None:
        mean, std = stats.get("audio_mean", x_t.device), stats.get("audio_std", x_t.device)
    x = ops.stft(x_t, nfft, int(hop_percent * nfft), mode=0, eps=eps, pad_at_end=True, fs=fs, mean=mean, std=std,
                 norm_eps=stats.eps if stats is not None else eps)
    if n_label_frames is not None and n_label_frames < x.shape[1]:      # "Reduce frames of audio" (:144-146)
        x = x[:, :n_label_frames].contiguous()
    return x


def process_utt(classifier, x_t, stats=None, n_label_frames=None, video=None, eps=EPS, std_norm=True):
    """One utterance through the reference's evaluator (``evaluate_audio_net.py:107-180``; with ``video`` (T,67,67) the AV
    variant ``evaluate_AV_net.py:148-250``): returns (y_hat_soft, y_hat_hard) on the CPU, shaped (1, T) like the
    reference's ``y_hat_soft[..., 0]``."""
    x = audio_features(x_t, stats, n_label_frames, eps=eps, std_norm=std_norm)
    lengths = [x.shape[1]]
    if video is None:
        y = classifier(x, lengths)
    else:
        v = video[None, :x.shape[1]].contiguous()
        y = classifier(x, stats.video(v) if (stats is not None and std_norm) else v, lengths)
    soft = torch.sigmoid(y[..., 0].detach().cpu())
    return soft, (soft > 0.5).int()


def evaluate_main(kind, make_model, checkpoint=None, waveform=False, n_items=16, out_dir="eval_out", wav_list=None,
                  stats=None, labels=None):
    """The body of ``scripts/evaluate_*_net.py``: per-utterance forward, sigmoid, threshold, save
    ``*_y_hat_soft.pt`` / ``*_y_hat_hard.pt`` (``evaluate_AV_net.py:236-250``); utterances are split across ranks
    (the reference's 4-process pool, ``:329-339``).

    ``wav_list`` (audio network): paths of 16 kHz utterances (.wav / .npz) that go through ``process_utt`` -- the
    reference's plumbing on real audio; ``labels`` optionally maps a path to its label tensor (frame count crop + saved
    next to the predictions for ``run_metrics``).  Without it a synthetic ragged data source stands in for the HDF5
    datasets."""
    rank, world, local = avd.init_from_env("nccl")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    model = make_model()
    if checkpoint:
        model.load_state_dict(torch.load(checkpoint, map_location="cpu", weights_only=True))
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.perf_counter()
    if wav_list is not None:
        if kind != "audio":
            raise ValueError("wav_list drives the audio evaluator (evaluate_audio_net.py); video needs the HDF5 readers")
        with torch.no_grad():
            for i in range(rank, len(wav_list), world):
                x_t, fs = load_waveform(wav_list[i])
                if fs != 16000:
                    raise ValueError("%s: expected 16 kHz audio, got %d Hz" % (wav_list[i], fs))
                y = labels.get(wav_list[i]) if labels else None
                soft, hard = process_utt(model, x_t.to(device), stats, None if y is None else y.shape[-1])
                base = os.path.join(out_dir, os.path.splitext(os.path.basename(wav_list[i]))[0])
                torch.save(hard, base + "_y_hat_hard.pt")
                torch.save(soft, base + "_y_hat_soft.pt")
                if y is not None:
                    torch.save(y.int().cpu(), base + "_label.pt")
    else:
        ds = SyntheticAV(n_items, kind, waveform=waveform, seed=7)
        collate = pick_collate(kind, waveform)
        with torch.no_grad():
            for i in range(rank, n_items, world):
                batch = collate([ds[i]])
                lengths, logits, y = forward_batch(model, kind, batch, device, waveform, stats)
                soft = torch.sigmoid(logits[..., 0].detach().cpu())                      # (1,T), evaluate_AV_net.py:236-240
                torch.save(soft, os.path.join(out_dir, "utt%04d_y_hat_soft.pt" % i))
                torch.save((soft > 0.5).int(), os.path.join(out_dir, "utt%04d_y_hat_hard.pt" % i))
                torch.save(y[..., 0].int().cpu(), os.path.join(out_dir, "utt%04d_label.pt" % i))   # synthetic stand-in for the labels
    if rank == 0:
        print("Finished in {:.2f} seconds".format(time.perf_counter() - t0))


def metrics_main(out_dir="eval_out", confidence=0.95, eps=1e-8):
    """The body of ``scripts/run_metrics_dnn_classif.py`` (``:102-300``) for the classifier outputs: per utterance
    ``f1_loss(y_hat_hard, y)`` -> (accuracy, precision, recall, F1), then ``compute_stats`` tables with Student-t
    confidence intervals.  Reads the ``*_y_hat_hard.pt`` / ``*_label.pt`` pairs that ``evaluate_main`` wrote."""
    import glob
    from packages.metrics import compute_stats
    from packages.models.utils import f1_loss
    rows = []
    for hard_path in sorted(glob.glob(os.path.join(out_dir, "*_y_hat_hard.pt"))):
        y_hat = torch.load(hard_path, weights_only=True).reshape(-1)
        y = torch.load(hard_path.replace("_y_hat_hard.pt", "_label.pt"), weights_only=True).reshape(-1)
        rows.append(tuple(float(v) for v in f1_loss(y_hat, y.long(), eps)))
    if len(rows) < 2:
        raise SystemExit("need at least two evaluated utterances in %s (run scripts/evaluate_*_net.py first)" % out_dir)
    return compute_stats(metrics_keys=["accuracy", "precision", "recall", "f1score"], all_metrics=rows, model_data_dir=out_dir,
                         confidence=confidence)
