// r6_policy_device.hpp -- device code of the fused actor-critic forward (SB3 1.6 ActorCriticPolicy.forward, the caller of
// Rocket6DOF.step in a rollout: main_6DOF.py:107-139), shared by r6::policy_act_kernel (r6_policy.hip) and the closed-loop
// rollout kernel (r6_rollout.hip).  One definition of the arithmetic, so that a fused rollout is bit for bit the sequence
// r6_policy_act -> r6_step.  See r6_policy.hip for the layout and the numerics.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/r6dof.h"
#include "r6_device.hpp"

namespace r6p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// x = hi + lo with hi, lo in half precision: 22 significant bits of each fp32 value survive
struct Split {
    h8 hi, lo;
};
__device__ __forceinline__ void split2(float a, float b, h2& hi, h2& lo) {
    hi = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
    lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a - (float)hi[0], b - (float)hi[1]));
}
__device__ __forceinline__ Split split8(const float (&x)[8]) {
    Split s;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        h2 hi, lo;
        split2(x[j], x[j + 1], hi, lo);
        s.hi[j] = hi[0];
        s.hi[j + 1] = hi[1];
        s.lo[j] = lo[0];
        s.lo[j + 1] = lo[1];
    }
    return s;
}
// acc += A * B to fp32 accuracy: three half-precision products (lo*lo is below fp32 resolution)
__device__ __forceinline__ f32x16 mfma3(const h8& ah, const h8& al, const Split& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.hi, acc, 0, 0, 0);
}

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// tanh(z) = 1 - 2 r,  r = 1 / (exp(2z) + 1).  The kernel only ever forms r (exp2, add, rcp: the two quarter-rate
// ops are the VALU floor of this kernel; absolute error ~1e-7, saturates cleanly): the scale 2/ln2 of the
// exponent is folded into the weights and bias of the layer that PRODUCES z, and the affine map 1 - 2 r into the
// layer that CONSUMES it,  W a + b = (b + rowsum W) - 2 W r  -- both done once, when the weights are staged.
constexpr float kExpScale = 2.8853900817779268f;   // 2 / ln 2
__device__ __forceinline__ float half_sigmoid(float z_scaled) {
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z_scaled) + 1.0f);
}

// LDS image, H1/H2 = hidden widths (multiples of 32), NET in {0 = pi, 1 = vf}.  Half-precision part
// (offsets in halves; every fragment = 64 lanes x 8 halves = one ds_read_b128 per lane), hi then lo copy:
//   w1  [NET][H1/32][64][8]              A of layer 1 (K = 16): lane l, elem j: W1[mb*32 + (l&31)][8(l>>5) + j]
//                                         (k = od carries the bias against an obs column of ones, the rest is 0)
//   w2  [NET][H2/32][H1/32][2][64][8]    A of layer 2: lane l, block b, step s, elem j:
//                                         W2[mb*32 + (l&31)][32b + row(8s + j, l>>5)]
// fp32 part (offsets in floats, after the half part):
//   b2  [NET][H2/32][2][16]              accumulator init of layer 2: b2[mb*32 + row(r,h)]
//   w3  [2 (h)][H2/32][16][4]            last layer: {pi_w3[0..2][32b + row(r,h)], vf_w3[0][32b + row(r,h)]}
template <int H1, int H2>
struct Image {
    static constexpr int kW1 = 2 * (H1 / 32) * 64 * 8;                    // halves per copy
    static constexpr int kW2 = 2 * (H2 / 32) * (H1 / 32) * 2 * 64 * 8;
    static constexpr int kHalves = 2 * (kW1 + kW2);                       // hi + lo
    static constexpr int oW1h = 0, oW1l = kW1, oW2h = 2 * kW1, oW2l = 2 * kW1 + kW2;
    static constexpr int kB2 = 2 * (H2 / 32) * 2 * 16;
    static constexpr int kW3 = 2 * (H2 / 32) * 16 * 4;
    static constexpr int oB2 = kHalves / 2, oW3 = oB2 + kB2, kTotalFloats = oW3 + kW3;   // float offsets
};


// ---- stage the weights: torch Linear layout [out][in] fp32 -> split halves in fragment order in LDS (no barrier here) ----
template <int H1, int H2, int OD>
__device__ __forceinline__ void stage_policy_image(const R6PolicyNet& net, float* lds, const int tid, const int nthreads) {
    using Im = Image<H1, H2>;
    constexpr int kBiasK = OD;   // the obs column that is set to 1 and multiplies b1
    _Float16* ldsh = reinterpret_cast<_Float16*>(lds);
    auto put_split = [&](int off_hi, int off_lo, int e, float v) __attribute__((always_inline)) {
        const _Float16 hi = (_Float16)v;
        ldsh[off_hi + e] = hi;
        ldsh[off_lo + e] = (_Float16)(v - (float)hi);
    };
#pragma unroll 8
    for (int e = tid; e < Im::kW1; e += nthreads) {
        const int j = e & 7, l = (e >> 3) & 63, rest = e >> 9;
        const int mb = rest % (H1 / 32), nt = rest / (H1 / 32);
        const int k = 8 * (l >> 5) + j;
        const int m = mb * 32 + (l & 31);
        if (nt && net.shared) continue;   // one trunk: the second network's image is never read
        const float* w = nt ? net.vf_w1 : net.pi_w1;
        const float* b = nt ? net.vf_b1 : net.pi_b1;
        put_split(Im::oW1h, Im::oW1l, e, kExpScale * ((k < OD) ? w[m * OD + k] : ((k == kBiasK) ? b[m] : 0.0f)));
    }
#pragma unroll 8
    for (int e = tid; e < Im::kW2; e += nthreads) {
        const int j = e & 7, l = (e >> 3) & 63, sstep = (e >> 9) & 1, rest = e >> 10;
        const int b = rest % (H1 / 32), mb = (rest / (H1 / 32)) % (H2 / 32), nt = rest / ((H1 / 32) * (H2 / 32));
        const int k = 32 * b + rowmap(8 * sstep + j, l >> 5);
        const int m = mb * 32 + (l & 31);
        if (nt && net.shared) continue;
        const float* w = nt ? net.vf_w2 : net.pi_w2;
        put_split(Im::oW2h, Im::oW2l, e, -2.0f * kExpScale * w[m * H1 + k]);
    }
    for (int e = tid; e < Im::kB2; e += nthreads) {
        const int r = e & 15, hh = (e >> 4) & 1, rest = e >> 5;
        const int mb = rest % (H2 / 32), nt = rest / (H2 / 32);
        if (nt && net.shared) continue;
        const float* b = nt ? net.vf_b2 : net.pi_b2;
        const float* w = nt ? net.vf_w2 : net.pi_w2;
        const int m = mb * 32 + rowmap(r, hh);
        float rs = 0.0f;
        for (int k = 0; k < H1; ++k) rs += w[m * H1 + k];
        lds[Im::oB2 + e] = kExpScale * (b[m] + rs);
    }
    for (int e = tid; e < Im::kW3; e += nthreads) {
        const int o = e & 3, r = (e >> 2) & 15, rest = e >> 6;
        const int b = rest % (H2 / 32), hh = rest / (H2 / 32);
        const int k = 32 * b + rowmap(r, hh);
        lds[Im::oW3 + e] = -2.0f * ((o < 3) ? net.pi_w3[o * H2 + k] : net.vf_w3[k]);
    }
}

// last-layer constants of a launch: b3 + rowsum W3 (the 1 of 1 - 2 r) and the log-std; uniform scalar loads
struct HeadConsts {
    float b3_0, b3_1, b3_2, b3_v, ls0, ls1, ls2;
};
template <int H2>
__device__ __forceinline__ HeadConsts load_head_consts(const R6PolicyNet& net) {
    HeadConsts c;
    c.b3_0 = net.pi_b3[0];
    c.b3_1 = net.pi_b3[1];
    c.b3_2 = net.pi_b3[2];
    c.b3_v = net.vf_b3[0];
    for (int k = 0; k < H2; ++k) {
        c.b3_0 += net.pi_w3[k];
        c.b3_1 += net.pi_w3[H2 + k];
        c.b3_2 += net.pi_w3[2 * H2 + k];
        c.b3_v += net.vf_w3[k];
    }
    c.ls0 = net.log_std[0];
    c.ls1 = net.log_std[1];
    c.ls2 = net.log_std[2];
    return c;
}

// Both MLPs (or the one trunk) for the 32 envs of a wave tile: xs = this lane's 8 obs features (k = s + 8 h, bias column = 1),
// split; part[] = this lane half's share of (mean0, mean1, mean2, value) -- the other half holds the other 16 rows of every block.
template <int H1, int H2>
__device__ __forceinline__ void policy_tile_forward(const R6PolicyNet& net, const float* lds, const Split& xs, const int lane, const int h,
                                                    float (&part)[4]) {
    using Im = Image<H1, H2>;
    const _Float16* ldsh = reinterpret_cast<const _Float16*>(lds);
    part[0] = part[1] = part[2] = part[3] = 0.0f;
    // one network at a time (rolled): the two share nothing but x[], and unrolled the scheduler
    // interleaves them and triples the live registers
    const int n_nets = net.shared ? 1 : 2;   // one trunk feeds both heads, or one MLP per head
#pragma unroll 1
    for (int nt = 0; nt < n_nets; ++nt) {
        // ---- layer 2 accumulators start at the bias -----------------------------------------------------------
        f32x16 a2[H2 / 32];
#pragma unroll
        for (int mb = 0; mb < H2 / 32; ++mb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *(const f32x4*)&lds[Im::oB2 + ((nt * (H2 / 32) + mb) * 2 + h) * 16 + q * 4];
                a2[mb][q * 4 + 0] = b[0];
                a2[mb][q * 4 + 1] = b[1];
                a2[mb][q * 4 + 2] = b[2];
                a2[mb][q * 4 + 3] = b[3];
            }
        }
        // ---- layer 1 one 32-feature block at a time (one K = 16 step, B = the split obs row), fed straight
        //      into layer 2 -- registers 8s..8s+7 of the block are K-step (b, s) -- so that a single block of
        //      layer-1 activations is live instead of H1/32 of them (64 registers less for H1 = 128) ---------
#pragma unroll
        for (int blk = 0; blk < H1 / 32; ++blk) {
            f32x16 a1;
            {
                const int f = ((nt * (H1 / 32) + blk) * 64 + lane) * 8;
                const h8 wh = *(const h8*)&ldsh[Im::oW1h + f], wl = *(const h8*)&ldsh[Im::oW1l + f];
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                a1 = mfma3(wh, wl, xs, z);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[r] = half_sigmoid(a1[r]);
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = a1[sk * 8 + j];
                const Split bs = split8(v);
                const int g = blk * 2 + sk;
#pragma unroll
                for (int mb = 0; mb < H2 / 32; ++mb) {
                    const int f = (((nt * (H2 / 32) + mb) * (H1 / 32) * 2 + g) * 64 + lane) * 8;
                    const h8 wh = *(const h8*)&ldsh[Im::oW2h + f], wl = *(const h8*)&ldsh[Im::oW2l + f];
                    a2[mb] = mfma3(wh, wl, bs, a2[mb]);
                }
                asm volatile("" ::: "memory");   // keeps the scheduler from hoisting every group's LDS reads up front
            }
            __builtin_amdgcn_sched_barrier(0);   // one block at a time: do not start the next block's layer 1 early
        }

        // ---- r of layer 2, then the last layer on the VALU: this half's 16 rows per block (one uniform
        //      branch per network, not per value) --------------------------------------------------------
#pragma unroll
        for (int mb = 0; mb < H2 / 32; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[mb][r] = half_sigmoid(a2[mb][r]);
        }
        if (net.shared) {          // both heads read the one trunk
#pragma unroll
            for (int mb = 0; mb < H2 / 32; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 w3 = *(const f32x4*)&lds[Im::oW3 + ((h * (H2 / 32) + mb) * 16 + r) * 4];
                    part[0] = fmaf(w3[0], a2[mb][r], part[0]);
                    part[1] = fmaf(w3[1], a2[mb][r], part[1]);
                    part[2] = fmaf(w3[2], a2[mb][r], part[2]);
                    part[3] = fmaf(w3[3], a2[mb][r], part[3]);
                }
            }
        } else if (nt == 0) {
#pragma unroll
            for (int mb = 0; mb < H2 / 32; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 w3 = *(const f32x4*)&lds[Im::oW3 + ((h * (H2 / 32) + mb) * 16 + r) * 4];
                    part[0] = fmaf(w3[0], a2[mb][r], part[0]);
                    part[1] = fmaf(w3[1], a2[mb][r], part[1]);
                    part[2] = fmaf(w3[2], a2[mb][r], part[2]);
                }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < H2 / 32; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    part[3] = fmaf(lds[Im::oW3 + ((h * (H2 / 32) + mb) * 16 + r) * 4 + 3], a2[mb][r], part[3]);
            }
        }
    }
}

// The heads of one env from the two lane halves' shares: (mean0, mean1, mean2, value)
struct Heads {
    float m0, m1, m2, v;
};
__device__ __forceinline__ Heads combine_halves(const float (&part)[4], const HeadConsts& c) {
    Heads o;
    o.m0 = part[0] + __shfl_xor(part[0], 32) + c.b3_0;
    o.m1 = part[1] + __shfl_xor(part[1], 32) + c.b3_1;
    o.m2 = part[2] + __shfl_xor(part[2], 32) + c.b3_2;
    o.v = part[3] + __shfl_xor(part[3], 32) + c.b3_v;
    return o;
}

// Diagonal Gaussian of SB3 (DiagGaussianDistribution): sample, log-probability; noise = Philox4x32-10 keyed by
// (seed, step_index, global env) + Box-Muller.  a[] = the raw sample (what the rollout buffer stores).
struct Sample {
    float a0, a1, a2, logp;
};
__device__ __forceinline__ Sample gaussian_sample(const Heads& m, const HeadConsts& c, const bool deterministic, const uint64_t seed,
                                                  const uint64_t step_index, const uint64_t gid) {
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
    if (!deterministic) {
        const r6::U4 w = r6::philox4x32_10(
            r6::U4{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)},
            (uint32_t)seed ^ 0x5851F42Du, (uint32_t)(seed >> 32) ^ 0x4C957F2Du);
        // Box-Muller on (0,1] uniforms; v_sin/v_cos take their argument in revolutions
        const float u0 = (float)((w.x >> 8) + 1u) * 5.9604644775390625e-08f;
        const float u1 = (float)(w.y >> 8) * 5.9604644775390625e-08f;
        const float u2 = (float)((w.z >> 8) + 1u) * 5.9604644775390625e-08f;
        const float u3 = (float)(w.w >> 8) * 5.9604644775390625e-08f;
        const float r01 = sqrtf(-2.0f * __logf(u0)), r23 = sqrtf(-2.0f * __logf(u2));
        e0 = r01 * __builtin_amdgcn_cosf(u1);
        e1 = r01 * __builtin_amdgcn_sinf(u1);
        e2 = r23 * __builtin_amdgcn_cosf(u3);
    }
    Sample s;
    s.a0 = fmaf(__expf(c.ls0), e0, m.m0);
    s.a1 = fmaf(__expf(c.ls1), e1, m.m1);
    s.a2 = fmaf(__expf(c.ls2), e2, m.m2);
    s.logp = -0.5f * (e0 * e0 + e1 * e1 + e2 * e2) - (c.ls0 + c.ls1 + c.ls2) - 2.756815599614018f;   // 1.5 ln(2 pi)
    return s;
}

}  // namespace r6p
