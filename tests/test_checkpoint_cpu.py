"""Checkpoint interop at the state-dict level: plain tensors only, the reference's key names, and a
pickled-module file (what SBL/utils.py:22-33 writes) is refused instead of being unpickled."""
import pickle

import pytest
import torch


def _model(seed):
    from sbl_for_multilingual_lip_reading_amd.transformer.decoder import Decoder
    from sbl_for_multilingual_lip_reading_amd.transformer.encoder import Encoder
    from sbl_for_multilingual_lip_reading_amd.transformer.transformer import Transformer
    torch.manual_seed(seed)
    return Transformer(Encoder(512, 1, 8, 64, 64, 512, 2048), Decoder(0, 1, 58, 512, 1, 8, 64, 64, 512, 2048), None)


def test_round_trip_and_frontend_only_file(tmp_path):
    from sbl_for_multilingual_lip_reading_amd import checkpoint
    a, b = _model(1), _model(2)
    assert any(not torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    checkpoint.save_checkpoint(tmp_path / "ck.pt", a, epoch=3, tag="unit")
    meta = checkpoint.load_checkpoint(tmp_path / "ck.pt", b)
    assert meta["epoch"] == 3 and meta["tag"] == "unit"
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    # what the reference itself would do with our file (SBL/train.py:92-103 with a state dict instead of a module)
    c = _model(3)
    c.load_state_dict(torch.load(tmp_path / "ck.pt", weights_only=True)["model_state_dict"])
    assert torch.equal(c.decoder.tgt_word_prj_l2r.weight, a.decoder.tgt_word_prj_l2r.weight)
    # frontend-only weights (`visual_frontend(pt)`, video_frontend.py:179-188): bare state dict of the Lipreading module
    torch.save({k: v.clone() for k, v in a.visual_frontend.state_dict().items()}, tmp_path / "fe.pt")
    d = _model(4)
    checkpoint.load_checkpoint(tmp_path / "fe.pt", d)
    assert torch.equal(d.visual_frontend.frontend3D[0].weight, a.visual_frontend.frontend3D[0].weight)
    assert not torch.equal(d.encoder.linear_in.weight, a.encoder.linear_in.weight)
    with pytest.raises(KeyError):
        sd = {k: v for k, v in a.state_dict().items() if not k.startswith("encoder.")}
        torch.save({"model_state_dict": sd}, tmp_path / "partial.pt")
        checkpoint.load_checkpoint(tmp_path / "partial.pt", _model(5))
    with pytest.raises(TypeError):
        checkpoint.save_checkpoint(tmp_path / "bad.pt", a, thing=object())


def test_torch_adam_state_round_trips(tmp_path):
    """The reference's optimizer (torch.optim.Adam inside TransformerOptimizer, SBL/train.py:75): its nested state dict is
    saved and restored, so a resume continues with the moments and the Noam step it stopped at."""
    from sbl_for_multilingual_lip_reading_amd import checkpoint
    from sbl_for_multilingual_lip_reading_amd.transformer.optimizer import TransformerOptimizer
    a = _model(1)
    opt = TransformerOptimizer(torch.optim.Adam(a.encoder.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9))
    for _ in range(3):
        opt.zero_grad()
        sum((p * p).sum() for p in a.encoder.parameters()).backward()
        opt.step()
    checkpoint.save_checkpoint(tmp_path / "ck.pt", a, opt)
    b = _model(1)
    opt_b = TransformerOptimizer(torch.optim.Adam(b.encoder.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9))
    checkpoint.load_checkpoint(tmp_path / "ck.pt", b, opt_b)
    assert opt_b.step_num == 3
    sa, sb = opt.optimizer.state_dict()["state"], opt_b.optimizer.state_dict()["state"]
    assert len(sa) == len(sb) > 0
    for k in sa:
        assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"])
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 3.0
    with pytest.raises(TypeError):
        checkpoint._to_cpu({"x": object()})


class _Evil:
    def __reduce__(self):
        return (print, ("a pickled object of a checkpoint ran code",))


def test_pickled_module_checkpoints_are_refused(tmp_path):
    from sbl_for_multilingual_lip_reading_amd import checkpoint
    with open(tmp_path / "ref_style.tar", "wb") as f:
        pickle.dump({"epoch": 1, "model": _Evil()}, f)
    with pytest.raises(Exception):
        checkpoint.load_checkpoint(tmp_path / "ref_style.tar", _model(1))


def test_flat_model_trainable_ranges():
    from sbl_for_multilingual_lip_reading_amd import dp
    m = _model(7)
    flat = dp.FlatModel(m)
    assert flat.trainable_ranges() == [(0, flat.numel)]
    for p in m.encoder.parameters():                     # README stage 2: freeze the encoder
        p.requires_grad = False
    r = flat.trainable_ranges()
    a, b = flat.ranges["encoder."]
    assert r == [(0, a), (b, flat.numel)]
    m.decoder.layer_first_l2r.pos_ffn.w_1.weight.requires_grad = False      # a tensor in the middle of the decoder segment
    assert len(flat.trainable_ranges()) == 3
