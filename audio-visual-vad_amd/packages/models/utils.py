"""Drop-in for the hot-path members of ``packages/models/utils.py``: ``binary_cross_entropy`` (``:108-113``),
``binary_cross_entropy_2classes`` (``:115-116``), ``f1_loss`` (``:164-203``), ``method3`` (``:36-55``),
``weights_init_normal`` (``:5-26``).  The VAE-era helpers
of that file are not on the path and are not provided."""
import torch

from avvad import ops


def weights_init_normal(m, mean=0.0, std=0.005):
    """Class-name dispatch as in the reference.  Its callers pass ``(name, parameter)`` tuples
    (``AV_Net.py:60-62``), whose class name is ``tuple`` -> nothing matches -> no-op; kept that way."""
    name = m.__class__.__name__
    for key, (mu, sd) in (("Linear", (mean, std)), ("Conv2d", (mean, std)), ("ConvTranspose2d", (mean, std)),
                          ("Norm", (1.0, 0.02)), ("lstm", (1.0, 0.02))):
        if key in name:
            m.weight.data.normal_(mu, sd)
            if getattr(m, "bias", None) is not None:
                m.bias.data.zero_()
            return


def binary_cross_entropy(r, x, eps):
    """-mean(x log(sigmoid(r)+eps) + (1-x) log(1-sigmoid(r)+eps)) over all elements of one sequence
    slice ``r`` (T, y_dim) of logits -- fused HIP loss kernel (value and gradient)."""
    r2 = r.reshape(1, r.shape[0], -1)
    return ops.masked_bce(r2, x.reshape(r2.shape), [r2.shape[1]], eps)


def binary_cross_entropy_2classes(r1, r2, x, eps):
    """-mean(sum(x log(r1+eps) + (1-x) log(r2+eps), dim=-1)) on two probability outputs (reference ``:115-116``,
    imported by ``scripts/train_video_net.py:18``) -- HIP loss kernel, value and both gradients."""
    return ops.Bce2ClassesFn.apply(r1, r2, x, eps)


def batch_binary_cross_entropy(logits, targets, lengths, eps):
    """The reference's per-sequence Python loop (``scripts/train_AV_net.py:298-301``) as ONE kernel:
    sum over sequences of the mean loss over each sequence's valid frames."""
    return ops.masked_bce(logits, targets, lengths, eps)


def method3(out, lengths):
    """Last valid step of every sequence; ``out`` is the padded (B,T,H) LSTM output here."""
    idx = torch.as_tensor(lengths, device=out.device).long() - 1
    return out[torch.arange(out.shape[0], device=out.device), idx]


def f1_loss(y_hat_hard, y, epsilon=1e-8):
    """accuracy, precision, recall, F1 of 1-D hard predictions (host-side metric, not differentiated)."""
    y_pred = y_hat_hard.detach()
    y_true = y.detach()
    assert y_true.ndim == 1 and y_pred.ndim in (1, 2)
    if y_pred.ndim == 2:
        y_pred = y_pred.argmax(dim=1)
    tp = (y_true * y_pred).sum().to(torch.float32)
    tn = ((1 - y_true) * (1 - y_pred)).sum().to(torch.float32)
    fp = ((1 - y_true) * y_pred).sum().to(torch.float32)
    fn = (y_true * (1 - y_pred)).sum().to(torch.float32)
    accuracy = (tp + tn) / (tp + tn + fp + fn + epsilon)
    precision = tp / (tp + fp + epsilon)
    recall = tp / (tp + fn + epsilon)
    f1 = 2 * (precision * recall) / (precision + recall + epsilon)
    return accuracy, precision, recall, f1


def batch_f1(y_hat_hard, y, lengths, epsilon=1e-8):
    """Vectorised form of the caller loop ``scripts/train_AV_net.py:318-334``: per-sequence metrics over the
    valid frames, averaged over the batch.  y_hat_hard / y: (B,T,1) or (B,T)."""
    B, T = y.shape[0], y.shape[1]
    yp = y_hat_hard.reshape(B, T).to(torch.float32)
    yt = y.reshape(B, T).to(torch.float32)
    m = (torch.arange(T, device=y.device)[None, :] < torch.as_tensor(lengths, device=y.device)[:, None]).float()
    tp = (yt * yp * m).sum(1)
    tn = ((1 - yt) * (1 - yp) * m).sum(1)
    fp = ((1 - yt) * yp * m).sum(1)
    fn = (yt * (1 - yp) * m).sum(1)
    acc = (tp + tn) / (tp + tn + fp + fn + epsilon)
    prec = tp / (tp + fp + epsilon)
    rec = tp / (tp + fn + epsilon)
    f1 = 2 * prec * rec / (prec + rec + epsilon)
    return acc.mean(), prec.mean(), rec.mean(), f1.mean()
