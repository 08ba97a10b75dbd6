"""Mirror of SBL_Multilingual_Lip_reading/transformer/loss.py."""
from ._env import config, ops

IGNORE_ID = config.IGNORE_ID


def cal_performance(pred, gold, smoothing=0.0):
    """Calculate cross entropy loss, apply label smoothing if needed (loss.py:7-24).
    Args:
        pred: N x T x C, score before softmax
        gold: N x T
    Returns (loss, n_correct).  n_correct is a python int like the reference's `.sum().item()` — that is the
    one host sync of this call; use cal_performance_device() to keep both values on the device.
    """
    loss, stats = cal_performance_device(pred, gold, smoothing)
    return loss, int(stats[2].item())


def cal_performance_device(pred, gold, smoothing=0.0):
    """Same as cal_performance but sync-free: returns (loss, stats) with stats = device float[3]
    (sum of row losses, #tokens with gold != IGNORE_ID, #correct)."""
    pred = pred.reshape(-1, pred.size(-1))
    gold = gold.contiguous().view(-1)
    return ops.SmoothedCEFn.apply(pred, gold, float(smoothing), IGNORE_ID)


def cal_loss(pred, gold, smoothing=0.0):
    """Label-smoothed CE (loss.py:27-52); smoothing == 0 is plain CE with ignore_index."""
    return cal_performance_device(pred, gold, smoothing)[0]
