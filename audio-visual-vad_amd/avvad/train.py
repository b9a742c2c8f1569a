"""Training / evaluation loops shared by the ``scripts/train_*.py`` and ``scripts/evaluate_*.py`` entry points.

They mirror the loop bodies of the reference (``scripts/train_AV_net.py:252-448``, ``train_audio_net.py:190-372``,
``train_video_net.py:182-319``, ``evaluate_AV_net.py:148-250``): per batch -- move to the GPU, standardise,
forward, per-sequence masked BCE summed over the batch, backward, Adam, per-sequence accuracy/precision/recall/F1
-- with the reference's Python loops over the batch replaced by single fused calls, ``nn.DataParallel`` replaced by
one process per GPU + bucketed RCCL all-reduce, and a synthetic data source (the reference's HDF5 / wav readers are
out of scope; h5py and torchaudio are not installed in this image)."""
import os
import time

import torch

from . import dist as avd
from .optim import FlatAdam

EPS = 1e-8


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


def forward_batch(model, kind, batch, device, waveform):
    lengths = batch[0].to(device)
    data = [t.to(device, non_blocking=True) for t in batch[1:]]
    y = data[-1]
    if kind == "audio":
        x = data[0].unsqueeze(1) if waveform else data[0]
        return lengths, model(x, lengths), y
    if kind == "video":
        return lengths, model(data[0], lengths), y
    a = data[0].unsqueeze(1) if waveform else data[0]
    return lengths, model(a, data[1], lengths), y


def run_epoch(model, kind, loader, device, waveform, opt=None, reducer=None, log=None, log_interval=10):
    from packages.models.utils import batch_binary_cross_entropy, batch_f1
    train = opt is not None
    model.train(train)
    tot = dict(loss=0.0, acc=0.0, prec=0.0, rec=0.0, f1=0.0, n=0)
    for i, batch in enumerate(loader):
        lengths, logits, y = forward_batch(model, kind, batch, device, waveform)
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
               freeze_features=False, out_dir=None):
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
    reducer = avd.BucketReducer(opt.params, opt.flat_grad, opt.offsets) if world > 1 else None
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
        tr = run_epoch(model, kind, mk(ds_train, True), device, waveform, opt, reducer, log)
        with torch.no_grad():
            va = run_epoch(model, kind, mk(ds_valid, False), device, waveform)
        log("====> Epoch: {:2d}  train loss {:.3f} f1 {:.3f} | valid loss {:.3f} f1 {:.3f} | {:.1f} s".format(
            epoch, tr["loss"], tr["f1"], va["loss"], va["f1"], time.perf_counter() - t0))
        if rank == 0:                       # same checkpoint naming as train_AV_net.py:443-448
            torch.save(model.state_dict(), os.path.join(out_dir, "Video_Net_epoch_{:03d}_vloss_{:.2f}.pt".format(epoch, va["loss"])))
    return model


def evaluate_main(kind, make_model, checkpoint=None, waveform=False, n_items=16, out_dir="eval_out"):
    """The body of ``scripts/evaluate_*_net.py``: per-utterance forward, sigmoid, threshold, save
    ``*_y_hat_soft.pt`` / ``*_y_hat_hard.pt`` (``evaluate_AV_net.py:236-250``); utterances are split across ranks
    (the reference's 4-process pool, ``:329-339``)."""
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
    ds = SyntheticAV(n_items, kind, waveform=waveform, seed=7)
    collate = pick_collate(kind, waveform)
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(rank, n_items, world):
            batch = collate([ds[i]])
            lengths, logits, y = forward_batch(model, kind, batch, device, waveform)
            soft = torch.sigmoid(logits[0])
            torch.save(soft.cpu(), os.path.join(out_dir, "utt%04d_y_hat_soft.pt" % i))
            torch.save((soft > 0.5).int().cpu(), os.path.join(out_dir, "utt%04d_y_hat_hard.pt" % i))
            torch.save(y[0].int().cpu(), os.path.join(out_dir, "utt%04d_label.pt" % i))   # synthetic stand-in for the dataset's labels
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
