"""Checkpoint interop.

The reference saves the pickled *module objects* (`SBL/utils.py:22-33`: torch.save({'model': model, 'optimizer':
optimizer, ...})) and reloads them with torch.load (`SBL/train.py:92-103`), which unpickles arbitrary code.  The
modules here keep the reference's 537 state-dict names and shapes, so interop happens at the state-dict level and
nothing is ever unpickled:

  * a reference checkpoint is converted once, in an environment that trusts it, with
        torch.save({'model_state_dict': ckpt['model'].state_dict(), 'epoch': ...}, path)
    (or, for `visual_frontend(pt)`-style frontend weights, `video_frontend.py:179-188`, the plain state dict) and
    loaded here with `load_checkpoint` (weights_only=True);
  * `save_checkpoint` writes the same plain-tensor format (+ the flat Adam moments when a FusedAdam is given), which
    the reference loads with `model.load_state_dict(torch.load(path)['model_state_dict'])`.
"""
import torch


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_cpu(v) for v in obj]
    if obj is None or isinstance(obj, (int, float, str, bool)):
        return obj
    raise TypeError("optimizer state holds a %s: only tensors, numbers, strings and containers of them are saved" % type(obj).__name__)


def save_checkpoint(path, model, optimizer=None, **meta):
    """model: Transformer (or any module of this package).  optimizer: TransformerOptimizer / FusedAdam / None."""
    m = model.module if hasattr(model, "module") else model          # nn.DataParallel wrapper, SBL/train.py:246-250
    out = {"model_state_dict": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}}
    if optimizer is not None:
        inner = getattr(optimizer, "optimizer", optimizer)
        out["step_num"] = int(getattr(optimizer, "step_num", 0))
        # FusedAdam: flat tensors and scalars; torch.optim.Adam (the reference's optimizer, SBL/train.py:75): nested
        # {'state': {idx: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} - plain containers of tensors and
        # numbers either way, which torch.load(weights_only=True) reads back
        out["optimizer_state"] = _to_cpu(inner.state_dict())
    for k, v in meta.items():
        if not isinstance(v, (int, float, str, bool)):
            raise TypeError("checkpoint metadata must be plain scalars/strings (got %s for %r)" % (type(v).__name__, k))
        out[k] = v
    torch.save(out, path)


def load_checkpoint(path, model, optimizer=None, strict=True):
    """Loads tensors only (weights_only=True): a pickled-module checkpoint of the reference is refused by torch.
    Accepts either {'model_state_dict': ...} or a bare state dict; `visual_frontend.`-less keys of a frontend-only
    file (`video_frontend.py:179-188`) are loaded into model.visual_frontend.  Returns the metadata dict."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck.get("model_state_dict", ck) if isinstance(ck, dict) else ck
    m = model.module if hasattr(model, "module") else model
    own = m.state_dict()
    if not any(k in own for k in sd) and hasattr(m, "visual_frontend") and all(("visual_frontend." + k) in own for k in sd):
        m.visual_frontend.load_state_dict(sd, strict=strict)
    else:
        with torch.no_grad():                       # copy in place: flat (dp.FlatModel) parameter views stay valid
            missing = [k for k in own if k not in sd]
            extra = [k for k in sd if k not in own]
            if strict and (missing or extra):
                raise KeyError("state dict mismatch: missing %s, unexpected %s" % (missing[:5], extra[:5]))
            for k, v in sd.items():
                if k in own:
                    if own[k].shape != v.shape:
                        raise ValueError("shape mismatch for %s: %s vs %s" % (k, tuple(own[k].shape), tuple(v.shape)))
                    own[k].copy_(v)
    if optimizer is not None and isinstance(ck, dict):
        if "step_num" in ck and hasattr(optimizer, "step_num"):
            optimizer.step_num = int(ck["step_num"])
        inner = getattr(optimizer, "optimizer", optimizer)
        if ck.get("optimizer_state") is not None:
            inner.load_state_dict(ck["optimizer_state"])
    return {k: v for k, v in ck.items() if k not in ("model_state_dict", "optimizer_state")} if isinstance(ck, dict) else {}
