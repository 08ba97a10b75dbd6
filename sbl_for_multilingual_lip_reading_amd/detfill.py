"""Deterministic, torch-RNG-free tensor fill shared by the oracle, the golden
generator, the tests and bench.py.

Every tensor is a pure function of (its state-dict name, its shape, a salt), so
no weights ever have to be shipped: the golden script fills the *reference*
modules with it inside the build container, and the GPU box fills the HIP-backed
modules with the same routine ("How goldens are made").

The generator is splitmix64 over the flat element index, seeded by an FNV-1a
hash of the name; numpy only.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str, salt: int) -> np.uint64:
    h = 0xCBF29CE484222325 ^ (salt * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, salt: int = 0) -> np.ndarray:
    """float32 array in [-1, 1), a function of (name, shape, salt) only."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _fnv1a(name, salt)
    bits = _splitmix64(idx) >> np.uint64(40)            # 24 random bits
    u = bits.astype(np.float64) * (2.0 / (1 << 24)) - 1.0
    return u.astype(np.float32).reshape(shape)


def normal(name: str, shape, salt: int = 0) -> np.ndarray:
    """float32 ~N(0,1) by Box-Muller over two uniform streams."""
    u1 = (uniform(name + "#a", shape, salt).astype(np.float64) + 1.0) * 0.5
    u2 = (uniform(name + "#b", shape, salt).astype(np.float64) + 1.0) * 0.5
    u1 = np.clip(u1, 2.0 ** -25, 1.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32)


# Named sets of per-tensor gains (substring of the state-dict key -> factor) applied on top of the rules of fill_value.
# "varied": with the plain fill the residual stream of the randomly initialised decoder is dominated by a component that
# is the same for every position and sample (12 near-uniform attention averages), so every logit row has the same arg-max and
# the token feedback loop (decoder.py:166-186) never changes what is fed.  A strong token embedding and weak sub-layer output
# projections keep the stream token-dominated: arg-max ids then differ across steps, samples and directions.
GAIN_SETS = {"varied": {"tgt_word_emb": 30.0, "attn.fc.weight": 0.3, "w_2.weight": 0.3}}


def _gain(name: str, gains) -> float:
    if not gains:
        return 1.0
    if isinstance(gains, str):
        gains = GAIN_SETS[gains]
    g = 1.0
    for pat, f in gains.items():
        if pat in name:
            g *= f
    return g


def fill_value(name: str, shape, salt: int = 0, gains=None) -> np.ndarray:
    """Value for one state-dict entry of the SBL model (any of its 537 keys).

    Rules (by key suffix / rank), chosen so train-mode BatchNorm and LayerNorm
    see non-trivial affine parameters and positive running variances:
      * rank >= 2 (conv kernels, Linear weights, embedding): xavier-uniform
        bound sqrt(6/(fan_in+fan_out)), the init the reference ends up with
        (SBL/transformer/transformer.py:18-20).
      * '*.running_var': 1 + 0.25*|u|;  '*.running_mean': 0.1*u
      * '*.num_batches_tracked': 0
      * BN / LN 'weight' (rank 1): 1 + 0.1*u;  rank-1 'bias': 0.05*u
    """
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    u = uniform(name, shape, salt)
    if len(shape) >= 2:
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        bound = np.sqrt(6.0 / (fan_in + fan_out))
        return (u * np.float32(bound * _gain(name, gains))).astype(np.float32)
    if name.endswith("running_var"):
        return (1.0 + 0.25 * np.abs(u)).astype(np.float32)
    if name.endswith("running_mean"):
        return (0.1 * u).astype(np.float32)
    if name.endswith("weight"):
        return (1.0 + 0.1 * u).astype(np.float32)
    return (0.05 * u).astype(np.float32)


def fill_state_dict(shapes: dict, salt: int = 0, gains=None) -> dict:
    """{name: shape} -> {name: np.ndarray}; 'pe' buffers are skipped (they are
    computed, not learned: SBL/transformer/module.py:17-24)."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith(".pe"):
            continue
        out[name] = fill_value(name, shape, salt, gains)
    return out


def synthetic_batch(B: int, T: int, H: int, W: int, salt: int = 7, max_tgt: int = 14,
                    vocab: int = 58):
    """Synthetic minibatch of SURVEY.md section 8(d): clips ~N(0,1) float32
    (B,T,H,W); per-sample target length U{3..14}, ids U{2..vocab-1}, int64
    (B,14) padded with -1; r2l = the reversed ids, same padding
    (cf. SBL/data_gen.py:297-302)."""
    x = normal("clips", (B, T, H, W), salt)
    lens = ((uniform("tgt_len", (B,), salt).astype(np.float64) + 1.0) * 0.5 * 12).astype(np.int64) + 3
    lens = np.clip(lens, 3, max_tgt)
    ids = ((uniform("tgt_ids", (B, max_tgt), salt).astype(np.float64) + 1.0) * 0.5 * (vocab - 2)).astype(np.int64) + 2
    ids = np.clip(ids, 2, vocab - 1)
    l2r = np.full((B, max_tgt), -1, dtype=np.int64)
    r2l = np.full((B, max_tgt), -1, dtype=np.int64)
    for b in range(B):
        n = int(lens[b])
        l2r[b, :n] = ids[b, :n]
        r2l[b, :n] = ids[b, :n][::-1]
    return x, l2r, r2l
