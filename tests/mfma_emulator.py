"""A numpy replay of the kernel's MFMA layer walk over the PACKED fragment stream
(test infrastructure).  It uses only the documented gfx950 lane/register maps
(cdna_hip_programming.md section 3):

  v_mfma_f32_32x32x16_{bf16,f16}: lane l (i=l&31, h=l>>5) holds A[i][8h+j], B[8h+j][i], j=0..7
  v_mfma_f32_32x32x2_f32       : lane l holds A[i][h], B[h][i]
  C/D (all of them)            : lane l reg r <-> D[(r&3)+8(r>>2)+4h][i]

and the kernel's conventions (mlp_core.hpp): an accumulator tile, converted
register-for-register, IS the next layer's B operand.  If the host packer's K
permutation were wrong, the replay would disagree with the oracle by O(1).
"""
import numpy as np

CHUNK_FRAGS = 16


def _row_of(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def bf16_round(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


class Stream:
    def __init__(self, raw: bytes, mode: str):
        self.mode = mode
        if mode == "f32":
            self.frags = np.frombuffer(raw, np.float32).reshape(-1, 64, 4)
        elif mode == "bf16":
            u = np.frombuffer(raw, np.uint16).reshape(-1, 64, 8).astype(np.uint32) << 16
            self.frags = u.view(np.float32)
        else:
            self.frags = np.frombuffer(raw, np.float16).reshape(-1, 64, 8).astype(np.float32)
        self.pos = 0

    def layer_start(self):
        self.pos = (self.pos + CHUNK_FRAGS - 1) // CHUNK_FRAGS * CHUNK_FRAGS

    def next(self):
        f = self.frags[self.pos]
        self.pos += 1
        return f


def tiles_from_matrix(X):
    """X (32*KT, 32 samples) fp32 -> per-lane register image regs[KT][64 lanes][16]."""
    KT = X.shape[0] // 32
    regs = np.zeros((KT, 64, 16), np.float32)
    for t in range(KT):
        for l in range(64):
            i, h = l & 31, l >> 5
            for r in range(16):
                regs[t, l, r] = X[32 * t + _row_of(r, h), i]
    return regs


def matrix_from_tiles(regs):
    KT = regs.shape[0]
    X = np.zeros((32 * KT, 32), np.float32)
    for t in range(KT):
        for l in range(64):
            i, h = l & 31, l >> 5
            for r in range(16):
                X[32 * t + _row_of(r, h), i] = regs[t, l, r]
    return X


def dense(stream: Stream, bias_rows, in_regs, MT, quant):
    """One layer.  in_regs[KT][64][16] are the operand tiles as the kernel holds them (already
    quantised to the mode's operand type).  Returns raw accumulators out[MT][64][16]."""
    mode = stream.mode
    x3 = mode == "f16x3"                         # in_regs = [hi, lo] operand images (quantize())
    KT = in_regs.shape[1] if x3 else in_regs.shape[0]
    SUB = 4 if mode in ("f32", "f16x3") else 2
    stream.layer_start()
    out = np.zeros((MT, 64, 16), np.float32)
    lanes = np.arange(64)
    li, lh = lanes & 31, lanes >> 5
    for m in range(MT):
        D = np.zeros((32, 32), np.float64)
        for l in range(64):                          # bias initialises the accumulator
            for r in range(16):
                D[_row_of(r, lh[l]), li[l]] = bias_rows[32 * m + _row_of(r, lh[l])]
        for t in range(KT):
            for s in range(SUB):
                frag = stream.next()
                if mode == "f32":
                    for e in range(4):               # one 32x32x2 MFMA per element
                        A = np.zeros((32, 2)); B = np.zeros((2, 32))
                        A[li, lh] = frag[lanes, e]
                        B[lh, li] = in_regs[t, lanes, 4 * s + e]
                        D += A @ B
                elif x3:
                    # split mode (mlp_core.hpp:ModeF16X3): fragment 2s' = W_hi -> X_hi and X_lo, fragment 2s'+1 = W_lo -> X_hi
                    A = np.zeros((32, 16)); Bh = np.zeros((16, 32)); Bl = np.zeros((16, 32))
                    for j in range(8):
                        A[li, 8 * lh + j] = frag[lanes, j]
                        Bh[8 * lh + j, li] = in_regs[0, t, lanes, 8 * (s >> 1) + j]
                        Bl[8 * lh + j, li] = in_regs[1, t, lanes, 8 * (s >> 1) + j]
                    D += A @ Bh
                    if s % 2 == 0:
                        D += A @ Bl
                else:
                    A = np.zeros((32, 16)); B = np.zeros((16, 32))
                    for j in range(8):
                        A[li, 8 * lh + j] = frag[lanes, j]
                        B[8 * lh + j, li] = in_regs[t, lanes, 8 * s + j]
                    D += A @ B
        for l in range(64):
            for r in range(16):
                out[m, l, r] = D[_row_of(r, lh[l]), li[l]]
    return out


def quantize(x, mode):
    if mode == "bf16":
        return bf16_round(x)
    if mode == "f16":
        return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
    if mode == "f16x3":
        x = np.clip(np.asarray(x, np.float32), -65504.0, 65504.0)
        hi = x.astype(np.float16).astype(np.float32)
        lo = (x - hi).astype(np.float16).astype(np.float32)
        return np.stack([hi, lo])
    return np.asarray(x, np.float32)
