"""TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the dropout RNG.

The reference draws dropout masks from TensorFlow's stateful RNG
(`tf.nn.dropout`, e.g. /root/reference/models/modules.py:15,27,69 and the
sites listed in SURVEY.md §8 a21).  TF's generator cannot be reproduced (TF is
not installed, version unpinned), so the build defines its own counter-based
generator - Philox4x32-7 (Salmon et al., SC'11: the fewest-round variant that passes
BigCrush; Random123 publishes known-answer vectors for 7 and for 10 rounds, both
checked in tests/test_oracle.py) - and uses it on both sides:
the HIP kernels (hual_amd/csrc/philox.h) and this numpy restatement.  With the
same (seed, offset) the oracle and the kernels drop exactly the same elements,
which makes parity tests with drop_rate > 0 exact instead of statistical.

Counter layout (must match hual_amd/csrc/philox.h):
    c0 = col >> 2      c1 = row      c2 = site id      c3 = offset (step)
    key = (seed & 0xffffffff, seed >> 32)
The four 32-bit outputs belong to columns 4*c0 + {0,1,2,3}.
An element is KEPT iff  out < thresh,  thresh = floor((1 - rate) * 2**32)
(clamped to 2**32-1) and kept elements are scaled by float32(1)/(float32(1)-float32(rate)).

The dropout on the attention probabilities (layers.py:86,91; modules.py:114 - half of all decisions of a step) draws
8 decisions from the eight 16-bit halves of ONE call (hual_amd/csrc/attn.hip): key k = 16 kt + 4 g + r of RNG row `rid` uses
the call with c0 = g + 4 (kt >> 1), output word 2 (kt & 1) + (r >> 1), half r & 1 (little endian) - the order in which a
lane of the forward kernel holds its scores; the element is KEPT iff half < t16 and kept elements are scaled by
1 / (1 - rate) exactly as tf.nn.dropout does (`mask_attn`; rounds 1-3 used 8-bit decisions with the scale 256 / t8).

The row-local sites INSIDE the network (every dropout of conv_block, dual_attn_block and the predictor's feature encoders
that is not an attention-probability site: SITE_CONV+l, SITE_DA+8li+{2,3,4}, SITE_FE+16p+{0..4,6,7,8}) draw 8 decisions
from the eight 16-bit halves of one call (hual_amd/csrc/tilecore.h drop_bits8_r): counter c0 = col >> 3, c1 = row; element
e = col & 7 uses word e >> 1, half e & 1 (little endian) and is KEPT iff half < t16, t16 = round(keep_prob * 65536) in
[1, 65536] (= (thresh + 2**15) >> 16); kept elements are scaled by 1 / (1 - rate) exactly as tf.nn.dropout does.  The keep
probability is t16 / 65536 (0.8000031 for rate 0.2 - closer to 1 - rate than the 2**-23 grid of TensorFlow's own float32
uniform draw allows it to be).  The clip-feature site SITE_VIDEO (model.py:47) and the four trilinear sites SITE_TRI+{0..3}
(ops.py:104; since round 4 - the Philox rounds were 30 % of the VALU time of the context-query forward) use the same 16-bit
decisions; the word / char embedding sites keep one decision per 32-bit word.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# ---- dropout call-site ids (mirrors include/hual_seqpan.h HUAL_SITE_*) -------------
SITE_WORD = 0        # modules.py:15   word_emb        rows = b*L+l,        cols = word_dim
SITE_CHAR = 1        # modules.py:27   char_emb        rows = (b*L+l)*C+c,  cols = char_dim
SITE_VIDEO = 2       # model.py:47     video_inputs    rows = b*T+t,        cols = vdim
SITE_CONV = 3        # +layer (0..3)   modules.py:69   shared conv_block, unified rows
SITE_DA = 8          # +8*li + {0 self probs, 1 cross probs, 2 dense_1 out, 3 LN2 out, 4 dense_2 out}
SITE_TRI = 24        # +{0 q2v.x1(v rows), 1 q2v.x2(q rows), 2 v2q.x1(q rows), 3 v2q.x2(v rows)}  ops.py:104
SITE_GUMBEL = 28     # layers.py:163-166 gumbel noise of the matching head: rows = b*T+t, the 4 words of ONE call = the 4 classes
SITE_FE = 32         # +16*pass + {0..3 conv layer, 4 LN1 out, 5 attn probs, 6 attn out, 7 LN2 out, 8 dense out}


PHILOX_ROUNDS = 7    # hual_amd/csrc/philox.h HUAL_PHILOX_ROUNDS


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=PHILOX_ROUNDS):
    """Vectorised Philox4x32-<rounds>.  All inputs broadcastable uint32-valued arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """the 10-round generator (Random123's default; the build's generator in rounds 1-3), kept for its known-answer test"""
    return philox4x32(c0, c1, c2, c3, k0, k1, rounds=10)


def keep_threshold(rate):
    """uint32 threshold: keep iff rnd < thresh.  Same double arithmetic as the C++ host code."""
    r = float(np.float32(rate))
    t = int(np.floor((1.0 - r) * 4294967296.0))
    return max(0, min(t, 0xFFFFFFFF))


def keep_scale(rate):
    return np.float32(1.0) / (np.float32(1.0) - np.float32(rate))


def keep_threshold16(rate):
    """16-bit threshold of the row-local sites inside the network: keep iff half < t16"""
    t = (keep_threshold(rate) + (1 << 15)) >> 16
    return max(1, min(t, 65536))


def uses_16bit_decisions(site):
    """the clip features, the trilinear sites + the row-local dropout sites of conv_block / dual_attn_block / feature_encoder
    (not the attention probabilities, which have their own lane order: mask_attn)"""
    if site == SITE_VIDEO or SITE_CONV <= site < SITE_CONV + 4:
        return True
    if SITE_DA <= site < SITE_TRI:
        return (site - SITE_DA) % 8 in (2, 3, 4)
    if SITE_TRI <= site < SITE_TRI + 4:
        return True
    if site >= SITE_FE:
        return (site - SITE_FE) % 16 in (0, 1, 2, 3, 4, 6, 7, 8)
    return False


class DropoutRNG:
    """mask(site, rows, ncols) -> float32 array [len(rows), ncols] of {0, scale}."""

    def __init__(self, seed, offset, rate):
        self.seed = int(seed)
        self.offset = int(offset)
        self.rate = float(rate)
        self.k0 = self.seed & 0xFFFFFFFF
        self.k1 = (self.seed >> 32) & 0xFFFFFFFF
        self.thresh = keep_threshold(rate)
        self.scale = keep_scale(rate)
        self.t16 = keep_threshold16(rate)

    def bits(self, site, rows, ncols):
        rows = np.asarray(rows, dtype=np.uint64).reshape(-1, 1)
        nblk = (ncols + 3) // 4
        c0 = np.arange(nblk, dtype=np.uint64).reshape(1, -1)
        o = philox4x32(c0, rows, np.uint64(site), np.uint64(self.offset & 0xFFFFFFFF), self.k0, self.k1)
        out = np.stack(o, axis=-1).reshape(rows.shape[0], nblk * 4)
        return out[:, :ncols]

    def mask16(self, site, rows, ncols):
        """16-bit decisions (module docstring): [len(rows), ncols] float32 of {0, 1 / (1 - rate)}"""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.uint64)).reshape(-1, 1)
        nblk = (ncols + 7) // 8
        c0 = np.arange(nblk, dtype=np.uint64).reshape(1, -1)
        o = philox4x32(c0, rows, np.uint64(site), np.uint64(self.offset & 0xFFFFFFFF), self.k0, self.k1)
        words = np.stack(o, axis=-1)                                   # [rows, call, word]
        halves = np.stack([words & np.uint32(0xFFFF), words >> np.uint32(16)], axis=-1)     # [rows, call, word, half]
        h = halves.reshape(rows.shape[0], nblk * 8)[:, :ncols]
        return np.where(h < np.uint32(self.t16), self.scale, np.float32(0.0)).astype(np.float32)

    def mask(self, site, rows, ncols):
        if self.rate == 0.0:
            return np.ones((len(np.atleast_1d(rows)), ncols), dtype=np.float32)
        if uses_16bit_decisions(int(site)):
            return self.mask16(site, rows, ncols)
        b = self.bits(site, rows, ncols)
        return np.where(b < np.uint32(self.thresh), self.scale, np.float32(0.0)).astype(np.float32)

    def mask_attn(self, site, rows, ncols):
        """16-bit decisions of the attention-probability sites in the forward kernel's lane order (module docstring):
        [len(rows), ncols] float32 of {0, 1 / (1 - rate)}."""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.uint64)).reshape(-1, 1)
        if self.rate == 0.0:
            return np.ones((rows.shape[0], ncols), dtype=np.float32)
        k = np.arange(ncols, dtype=np.int64)
        kt, g, r = k >> 4, (k >> 2) & 3, k & 3
        ncall = 4 * ((int(kt.max()) >> 1) + 1)
        c0 = np.arange(ncall, dtype=np.uint64).reshape(1, -1)
        o = philox4x32(c0, rows, np.uint64(site), np.uint64(self.offset & 0xFFFFFFFF), self.k0, self.k1)
        words = np.stack(o, axis=-1)                                   # [rows, call, word]
        w = words[:, g + 4 * (kt >> 1), 2 * (kt & 1) + (r >> 1)]       # [rows, ncols]
        half = (w >> (16 * (r & 1)).astype(np.uint32)) & np.uint32(0xFFFF)
        return np.where(half < np.uint32(self.t16), self.scale, np.float32(0.0)).astype(np.float32)


def gumbel_uniform(seed, offset, rows):
    """the uniform draws behind the gumbel noise of matching_loss (layers.py:163-166, ops.py:6-9): float32 [len(rows), 4] in
    [0, 1) on the 2**-24 grid - the top 24 bits of the four output words of the call (c0 = 0, c1 = row, c2 = SITE_GUMBEL,
    c3 = offset); hual_amd/csrc/heads.hip match_fwd_body draws the same numbers."""
    seed = int(seed)
    rows = np.atleast_1d(np.asarray(rows, dtype=np.uint64))
    o = philox4x32(np.uint64(0), rows, np.uint64(SITE_GUMBEL), np.uint64(int(offset) & 0xFFFFFFFF), seed & 0xFFFFFFFF,
                   (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(o, axis=-1)
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
