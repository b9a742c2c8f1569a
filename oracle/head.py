"""Oracle: packed multi-layer LSTM + Linear head, masked BCE, F1 -- CPU fp32.

Restates, with explicit time loops instead of cuDNN/ATen fused RNN calls:
  * ``pack_padded_sequence -> nn.LSTM(num_layers) -> pad_packed_sequence(total_length=T)
    -> nn.Linear``   ``packages/models/Audio_Net.py:50-60``, ``Video_Net.py:102-116``,
    ``AV_Net.py:127-140``
  * ``method3`` (last valid step of each sequence) ``packages/models/utils.py:36-55``
  * ``binary_cross_entropy`` ``packages/models/utils.py:108-113`` and its caller's
    per-sequence sum ``scripts/train_AV_net.py:298-301``
  * ``f1_loss`` ``packages/models/utils.py:164-203``

Packed semantics restated as masking: a sequence's (h, c) stop updating at its
length and padded output steps are ZERO (so the Linear there returns its bias).

Test infrastructure only (see ``oracle/__init__.py``).
"""
import torch


def lstm_layer(x, lengths, w_ih, w_hh, b_ih, b_hh):
    """x (B,T,In) -> (B,T,H); torch gate order i,f,g,o (nn.LSTM docs)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    lengths = torch.as_tensor(lengths)
    outs = []
    for t in range(T):
        gates = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = gates.chunk(4, dim=1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c_new = f * c + i * g
        h_new = o * torch.tanh(c_new)
        m = (t < lengths).to(x.dtype)[:, None]
        c = m * c_new + (1 - m) * c
        h = m * h_new + (1 - m) * h
        outs.append(m * h_new)
    return torch.stack(outs, dim=1)


def lstm_stack(x, lengths, sd, prefix, num_layers):
    """``nn.LSTM(num_layers=L)`` over a padded batch with packed semantics.
    ``sd[prefix + 'weight_ih_l0']`` etc. are the reference's state_dict keys."""
    y = x
    for l in range(num_layers):
        y = lstm_layer(y, lengths, sd["%sweight_ih_l%d" % (prefix, l)], sd["%sweight_hh_l%d" % (prefix, l)],
                       sd["%sbias_ih_l%d" % (prefix, l)], sd["%sbias_hh_l%d" % (prefix, l)])
    return y


def last_valid(y, lengths):
    """``method3`` (``models/utils.py:36-55``): y (B,T,H) -> (B,H) at t = len-1."""
    idx = (torch.as_tensor(lengths) - 1).long()
    return y[torch.arange(y.shape[0]), idx]


def linear(y, w, b):
    return y @ w.t() + b


def bce_with_eps(r, x, eps):
    """``binary_cross_entropy(r, x, eps)`` (``models/utils.py:108-113``): r logits."""
    s = torch.sigmoid(r)
    return -torch.mean(x * torch.log(s + eps) + (1 - x) * torch.log(1 - s + eps))


def bce_2classes(r1, r2, x, eps):
    """``binary_cross_entropy_2classes`` (``models/utils.py:115-116``): r1, r2 probabilities."""
    return -torch.mean(torch.sum(x * torch.log(r1 + eps) + (1 - x) * torch.log(r2 + eps), dim=-1))


def batch_loss(logits, targets, lengths, eps):
    """Caller loop ``scripts/train_AV_net.py:298-301``: per-sequence mean over the
    valid frames (and y_dim), SUMMED over the batch (the /B is commented out, ``:302``)."""
    loss = logits.new_zeros(())
    for b, n in enumerate([int(v) for v in lengths]):
        loss = loss + bce_with_eps(logits[b, :n], targets[b, :n].to(logits.dtype), eps)
    return loss


def f1_scores(y_hat_hard, y, epsilon=1e-8):
    """``f1_loss`` (``models/utils.py:164-203``): 1-D int predictions/targets ->
    (accuracy, precision, recall, f1)."""
    y_pred = y_hat_hard.detach()
    y_true = y.detach()
    tp = (y_true * y_pred).sum().to(torch.float32)
    tn = ((1 - y_true) * (1 - y_pred)).sum().to(torch.float32)
    fp = ((1 - y_true) * y_pred).sum().to(torch.float32)
    fn = (y_true * (1 - y_pred)).sum().to(torch.float32)
    accuracy = (tp + tn) / (tp + tn + fp + fn + epsilon)
    precision = tp / (tp + fp + epsilon)
    recall = tp / (tp + fn + epsilon)
    f1 = 2 * (precision * recall) / (precision + recall + epsilon)
    return accuracy, precision, recall, f1
