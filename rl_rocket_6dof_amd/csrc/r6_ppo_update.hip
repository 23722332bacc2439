// Fused PPO minibatch gradient: forward, loss and the WHOLE backward pass of the
// actor-critic in one launch -- parameter gradients come out, no activation ever goes to HBM.
//
// What it replaces: one minibatch step of SB3 1.6 `PPO.train` (stable_baselines3/ppo/ppo.py; the reference trains
// through it, main_6DOF.py:107-139) on the topology the reference uses, `policy_kwargs={'net_arch': [128, 64]}` =
// ONE shared tanh trunk obs -> 128 -> 64 with a 3-output action head and a scalar value head (R6PolicyNet.shared = 1):
//     policy.evaluate_actions(obs[mb], actions[mb]) ; ratio / clipped surrogate / value MSE / entropy ; loss.backward()
// In torch (ppo.py of this package) that is 5 row gathers, 4 GEMMs forward, 7 GEMMs backward -- the three tall
// weight-gradient products (64 x 65536)(65536 x 128) are the expensive ones -- and ~40 elementwise / reduction kernels:
// 0.52 ms of GPU time for 65 536 samples.  Here a wave owns a tile of 32 samples (the MFMA column block):
//
//   forward   z1 = W1 x, a1 = tanh z1, z2 = W2 a1, a2 = tanh z2 on the matrix cores, out^T[feature][sample] = W[feature][k] *
//             act^T[k][sample] as in r6_policy.hip: a layer's accumulator block IS the next layer's B operand; f16 hi + lo splits
//             of both operands with fp32 accumulation by default, v_mfma_f32_32x32x2_f32 (exact fp32 products) with
//             R6U_F16_FWD=0 -- the accuracy of both against float64 autograd is recorded below; heads (4 outputs) on the VALU;
//   loss      per sample, the formulas of r6_ppo_loss_grad -> g = d loss / d (mean0, mean1, mean2, value);
//   backward  delta2 = (W3^T g) (1 - a2^2) on the VALU; delta1 = (W2^T delta2) (1 - a1^2) on the matrix cores through
//             a W2^T image, delta2's accumulator-layout registers being the B operand again;
//   weight gradients  dW2 = delta2 a1^T, dW1 = delta1 x^T, dW3 = g a2^T, db2 = delta2 1: products that CONTRACT OVER
//             THE SAMPLES, i.e. over the lane dimension of everything above.  The tiles are transposed through a per-wave
//             LDS buffer ([feature][sample], pitch 36) and read back as MFMA operands with the sample index on K; the
//             sums live in accumulator registers for the whole launch: dW2 in 8 blocks of 32 x 32 (128 registers), dW1 + db1
//             (14 + 1 useful columns) in 8 blocks of 16 x 16 through v_mfma_f32_16x16x4_f32 (32 registers).
//             dW3 and db2 (4 + 1 useful columns) are summed on the VALU instead, lane = layer-2 feature = a row of the tile.
// After its last tile a workgroup adds its waves' blocks in LDS and writes one partial set; a second small launch sums
// the partial sets of all workgroups and scatters them into torch's parameter layouts.
//
// Tested against autograd of the torch modules (tests/test_gpu_ppo.py): every gradient within 2e-5 of its tensor's
// largest entry, loss statistics to 1e-5.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/r6dof.h"

extern "C" __attribute__((visibility("hidden"))) void r6_internal_set_error(const char* what);

namespace r6u {

constexpr int kBlock = 256, kWaves = 4;
constexpr int kPitch = 36;                 // floats per row of the transposed tiles: 32 samples (even ones first, then the odd ones) + 4,
                                           // 16-byte aligned rows whose 16-float halves are read conflict-free as 4 x ds_read_b128
[[maybe_unused]] constexpr int kP2 = 20;   // floats per lane of the layer-2 weight images (16 used): a conflict-free ds_read_b128 pitch
constexpr int kNumBlocks = 12;             // 32x32 slots of a partial set: dW2 (8) + dW1 (4; accumulated as two 16x16 blocks each, left half of the
                                           // slot), shared trunk [128, 64] or two MLPs (64, 64)
constexpr int kFeatNet = 5 * 64;           // per net and layer-2 feature f: dW3^T[f][0..3] and db2[f], summed on the VALU with lane = feature
constexpr int kFeat = 2 * kFeatNet;        // room for two nets (a shared trunk uses the first half)
constexpr int kScalars = 16;               // db3[4], dlog_std[3], pg, v, clip_frac, kl (+ pad)
constexpr int kOffFeat = kNumBlocks * 1024, kOffScal = kOffFeat + kFeat;
constexpr int kPartial = kOffScal + kScalars;   // one workgroup's partial set: [blocks][feat][scalars]
#ifdef R6U_STAMPS
constexpr int kStampSlots = 20;
#ifndef R6U_STAMP_SET
#define R6U_STAMP_SET 0   // 1: the staging in four parts, 2: the end-of-launch reduction in four parts (scripts/ppo_grad_fixed_cost.py --set=N)
#endif
#endif
constexpr float kExpScale = 2.8853900817779268f;   // 2 / ln 2

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Every layer product of the kernel except dW1 can run as v_mfma_f32_32x32x16_f16 on f16 hi + lo splits of both operands (three or
// four products per K-step of 16, fp32 accumulation; the scheme of r6_policy.hip) instead of v_mfma_f32_32x32x2_f32 -- 1/16 of the
// matrix time per K for 3-4x the instructions:
//   R6U_F16_FWD   forward layers 1 and 2 (B operand = an activation block in accumulator layout);
//   R6U_F16_BWD   the delta1 chains; delta2 is scaled per tile by a power of two first (gradients carry 1 / n and have any magnitude);
//   R6U_F16_DW2   dW2 = delta2 a1^T (contraction over the samples), the scaled delta2 again: a tile's products are summed in a block
//                 of their own and join the launch's sums times the inverse factor.
// The split keeps hi = the operand rounded towards zero to f16 and lo = the remainder ROUNDED TO NEAREST (R6U_SPLIT_LO_RN): 22-23 of
// the 24 bits, unbiased.  With a truncated lo (round 2's split, one v_cvt_pkrtz for both halves) every product errs the same way and
// sums over the samples keep the bias: the value head's bias gradient, which cancels to a few per cent of its terms, then missed
// the autograd test's 2e-5 by 1.4x; with the rounded lo the f16 forms are as close to FLOAT64 autograd as the exact-fp32 kernel and as
// torch's own fp32 autograd (scripts/ppo_grad_accuracy.py: worst gradient entry 0.5-3.6e-6 of its tensor's largest, torch fp32
// 0.5e-6-1.5e-5, exact-fp32 kernel 0.5-6.7e-6).  Measured on MI355X (scripts/ppo_grad_ab.py, 65 536 samples, gradient + reduce
// launches): exact fp32 everywhere 63.4 us; delta1 chains in f16 58.2; + this round's staging / end-of-launch reduction 50.6;
// + f16 forward 45.9 (all four partial products; three: 45.0, the default -- lo * lo is below what the split itself keeps);
// + f16 dW2 42.6 us with four, 41.8 with three forward products (two (64, 64) MLPs: 67.4 -> 50.3).  dW1 (16x16x4 fp32, 2048 cycles of a tile) stays exact:
// delta1 has no cheap bound for an f16 scale.  All three default to 1; =0 gives the exact-fp32 forms back.
#ifndef R6U_F16_FWD
#define R6U_F16_FWD 1
#endif
#ifndef R6U_F16_BWD
#define R6U_F16_BWD 1
#endif
#define R6U_F16_ANY (R6U_F16_FWD || R6U_F16_BWD)
#ifndef R6U_SPLIT_LO_RN
#define R6U_SPLIT_LO_RN 1
#endif
#ifndef R6U_F16_DW2
#define R6U_F16_DW2 R6U_F16_BWD
#endif
#if R6U_F16_DW2 && !R6U_F16_BWD
#error "R6U_F16_DW2 uses the delta-chain scaling of R6U_F16_BWD"
#endif
// R6U_DW2_DIRECT (with R6U_F16_DW2): dW2's f16 products accumulate straight into the launch's sums.  delta2^T, dW2's A operand, is
// scaled by ONE power of two per wave and net for the whole launch -- 2^(13 - e_base), e_base = floor(log2 max|delta2|) of the
// wave's first non-zero tile + R6U_DW2_BASE_SHIFT -- instead of the tile's own factor, so every tile's products carry the same
// factor and need neither a block of their own nor the join (16 x 8 fused multiply-adds per tile plus the copies between the
// register files around them).  The factor leaves the sums once, after the tile loop.  Nothing in the tile loop branches on it: a
// branch in the block loop splits its scheduling regions, and the allocator then spills a few hundred registers (tried: rescaling
// the sums when a tile outgrows the factor; moving the excess to the a1 operand).  Instead the first tile's largest entry is placed
// low, at [2^-3, 2^-2) -- f16 keeps 2^18 above it, and below it a hi + lo pair resolves 2^-24 absolute = 2^-21 of that entry
// (subnormal halves: the matrix pipe keeps them; scripts/ppo_grad_accuracy.py: the same errors as with the entry at 2^13) --
// and a tile more than 2^18 times the wave's first one saturates in this product (MODE.FP16_OVFL: +-65504, finite;
// tests/test_gpu_ppo.py::test_fused_minibatch_grad_with_tiles_of_very_different_size).  The delta1 chains keep the tile's own
// factor (their B operand, d2s).  =0 (default): a block per tile and the join.
// MEASURED, NOT THE DEFAULT (round 5, scripts/jobs/r5_job39.sh - r5_job42.sh): the tile loop loses 200 of 3188 instructions and the
// launch 0.2-0.5 us of 36 (65 536 samples), 1 us of 314 (2^20): stage E of the block loop, where the join ran, is bound by dW1's 16
// fp32 matrix instructions once the join is gone, not by the vector work.  (A first timing build without the factor logic read
// -12 us at 2^20: its sums ran to infinity, and the chip clocks higher on such data.)  Same accuracy as the default
// (scripts/ppo_grad_accuracy.py), but a saturation regime the default does not have: off.
#ifndef R6U_DW2_DIRECT
#define R6U_DW2_DIRECT 0
#endif
#ifndef R6U_DW2_BASE_SHIFT
#define R6U_DW2_BASE_SHIFT 16   // the first tile's largest delta2 entry goes to [2^(13 - shift), 2^(14 - shift)) = [2^-3, 2^-2): 2^18 of headroom above it, and
                                // a hi + lo pair still resolves 2^-24 (f16 subnormals: the matrix pipe keeps them) = 2^-21 of that entry
#endif
#if R6U_DW2_DIRECT && !R6U_F16_DW2
#error "R6U_DW2_DIRECT is a form of the f16 dW2 products"
#endif
#define R6U_FAST_STAGING R6U_F16_BWD   // images with f16 delta-chain fragments (either forward form): staged with every load in flight at once
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
struct Split {
    h8 hi, lo;
};
__device__ __forceinline__ void split2(float a, float b, h2& hi, h2& lo) {
    hi = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
#if R6U_SPLIT_LO_RN
    lo[0] = (_Float16)(a - (float)hi[0]);   // round to nearest: a truncated lo term biases every product the same way, and sums over the samples keep the bias
    lo[1] = (_Float16)(b - (float)hi[1]);
#else
    lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a - (float)hi[0], b - (float)hi[1]));
#endif
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
// acc += A * B with all four partial products (-DR6U_FWD_MFMA=mfma4: the forward's error reaches every gradient through the loss,
// and lo * lo, 2^-22 of a product, is the largest term the three-product form drops; the fourth f16 MFMA costs 32 cycles)
__device__ __forceinline__ f32x16 mfma4(const h8& ah, const h8& al, const Split& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.hi, acc, 0, 0, 0);
}
// acc += A * B to fp32 accuracy: three half-precision products (lo * lo is below fp32 resolution)
__device__ __forceinline__ f32x16 mfma3(const h8& ah, const h8& al, const Split& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b.hi, acc, 0, 0, 0);
}

#ifndef R6U_SCHED
#define R6U_SCHED 56   // software pipelines: 8 = layer 2 of the forward, 16 = the block loop of the backward (shared trunk only), 32 = layer 1; 1, 2, 4: dev experiments
#endif
#ifndef R6U_ABLATE
#define R6U_ABLATE 0   // dev: leave out 1 = the dW3 sums, 2 = the dW1 products, 4 = the db2 row sums (timing only: wrong gradients)
#endif
#ifndef R6U_FWD_MFMA
#define R6U_FWD_MFMA mfma3   // mfma4 adds lo * lo: 0.8 us per 65 536-sample launch for no measurable accuracy (scripts/ppo_grad_accuracy.py)
#endif
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// 16 x 16 output, K = 4 per instruction (lane l: A row / B column l % 16, k = l / 16; D register r: row 4 (l / 16) + r, column l % 16):
// the shape of dW1 = delta1 x^T, whose 14 + 1 useful columns fill a 16-wide block where the 32-wide one was half padding
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// tanh of an accumulator block, stage by stage over groups of eight: a value that comes off the transcendental unit cannot be
// used by the very next instruction (the compiler pads with s_nop), and element by element -- the order the scheduler kept for
// a block that sits in accumulation registers -- every exp and every rcp was followed by such a bubble (192 per tile)
__device__ __forceinline__ void tanh_half(const f32x16& z, f32x16& a, const int g8) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = kExpScale * z[g8 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(e[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] += 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[g8 + j] = fmaf(-2.0f, e[j], 1.0f);
}
__device__ __forceinline__ void tanh_block(const f32x16& z, f32x16& a) {
    tanh_half(z, a, 0);
    __builtin_amdgcn_sched_barrier(0);
    tanh_half(z, a, 8);
    __builtin_amdgcn_sched_barrier(0);
}
// tanh(z) = 1 - 2 / (exp(2z) + 1): v_exp_f32 + v_rcp_f32, absolute error ~1e-7, saturates cleanly
__device__ __forceinline__ float tanh_fast(float z) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(kExpScale * z) + 1.0f), 1.0f);
}
// 16 (8) consecutive floats of LDS as 4 (2) ds_read_b128: one MFMA chain's operands are fetched before the chain starts,
// so the matrix core does not wait for an LDS round trip per step (one wave per SIMD: nobody else would hide it)
template <int N>
__device__ __forceinline__ void ldv(const float* p, float (&v)[N]) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
        v[4 * q + 0] = t[0];
        v[4 * q + 1] = t[1];
        v[4 * q + 2] = t[2];
        v[4 * q + 3] = t[3];
    }
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS image of the weights (floats): per lane the values of one MFMA chain are contiguous (H1, H2 multiples of 32):
//   w1   [H1/32][64][kP1]            A of layer 1: lane l, step s: W1[blk*32 + (l&31)][k = 2s + (l>>5)]; k == OD carries b1
//   w2   [H2/32][H1/32][64][kP2]     A of layer 2: lane l, step r: W2[mb*32 + (l&31)][32b + row(r, l>>5)]
//   w2t  [H1/32][H2/32][64][kP2]     A of the delta chain: W2[32c + row(r, l>>5)][ib*32 + (l&31)]
//   b2   [H2/32][2][16]              accumulator init of layer 2: b2[mb*32 + row(r,h)]
//   w3   [2][H2/32][16][4]           heads: {pi_w3[0..2][32c + row(r,h)], vf_w3[32c + row(r,h)]}
// NETS = 1: one trunk feeding both heads (R6PolicyNet.shared); NETS = 2: net 0 = the policy MLP, net 1 = the value MLP, every
// image holds net 0's chains first.  w3 rows are 4 wide for both nets (columns 0..2 action head, 3 value head), with zeros in
// the columns a net does not feed, so heads, delta2 and the dW3 sums are the same code for every net.
template <int NETS, int H1, int H2>
struct Image {
    static constexpr int kP1 = (NETS == 2) ? 8 : 12;   // floats per lane of the layer-1 image (8 used); 12 is the conflict-free pitch, two
                                                       // nets trade it (2 reads per chain, 2-way conflicts) for 4 KB of LDS
    // f16 images (offsets and sizes still in floats = pairs of halves): per chain and K-step of 16 one fragment of 64 lanes x 8
    // halves (one ds_read_b128 per lane), a hi copy followed by a lo copy:
    //   w1   [chain = net*B1 + blk][hi|lo][64][8]                  lane l, elem j: W1[blk*32 + (l&31)][k = 8(l>>5) + j]; k == OD carries b1
    //   w2   [chain = (net*B2 + mb)*B1 + b][step s][hi|lo][64][8]   W2[mb*32 + (l&31)][32b + row(8s + j, l>>5)]
    //   w2t  [chain = (net*B1 + ib)*B2 + c][step s][hi|lo][64][8]   W2[32c + row(8s + j, l>>5)][ib*32 + (l&31)]
    static constexpr int kFrag = 64 * 8 / 2;           // floats per fragment copy
#if R6U_F16_FWD
    static constexpr int oW1 = 0, kW1 = NETS * (H1 / 32) * 2 * kFrag;
    static constexpr int oW2 = oW1 + kW1, kW2 = NETS * (H2 / 32) * (H1 / 32) * 2 * 2 * kFrag;
#else
    static constexpr int oW1 = 0, kW1 = NETS * (H1 / 32) * 64 * kP1;
    static constexpr int oW2 = oW1 + kW1, kW2 = NETS * (H2 / 32) * (H1 / 32) * 64 * kP2;
#endif
    static constexpr int oW2T = oW2 + kW2;
#if R6U_F16_BWD
    static constexpr int kW2T = NETS * (H2 / 32) * (H1 / 32) * 2 * 2 * kFrag;
#else
    static constexpr int kW2T = NETS * (H2 / 32) * (H1 / 32) * 64 * kP2;
#endif
#if R6U_F16_ANY
    // the 11 running sums of a sample-owning lane (lane half 0): [k][wave][32] -- the f32 images' padding held them in round 2
    static constexpr int oStat = oW2T + kW2T, kStat = 11 * kWaves * 32;
    static constexpr int oB2 = oStat + kStat, kB2 = NETS * (H2 / 32) * 2 * 16;
#else
    static constexpr int oB2 = oW2T + kW2T, kB2 = NETS * (H2 / 32) * 2 * 16;
#endif
    static constexpr int oW3 = oB2 + kB2, kW3 = NETS * 2 * (H2 / 32) * 16 * 4;
    static constexpr int kImage = oW3 + kW3;
    // per-wave staging (transposed tiles): a2 / delta2 [H2][kPitch], one 32-row block [32][kPitch], x [16][kPitch], g [4][kPitch]
    static constexpr int sT2 = 0, sBlk = H2 * kPitch, sX = sBlk + 32 * kPitch, sG = sX + 16 * kPitch;
    static constexpr int kStage = sG + 4 * kPitch;
    static constexpr int kTotal = kImage + kWaves * kStage;
    static_assert(kTotal * 4 <= 160 * 1024, "LDS budget of one workgroup per CU");
};

struct Args {
    const float* obs;
    const float* action;
    const float* old_logp;
    const float* adv;
    const float* ret;
    const int64_t* index;   // minibatch rows (NULL: rows 0..n-1)
    const float* adv_norm;  // [mean, 1 / (std + 1e-8)] or NULL
    int64_t n;
    float clip, vf_coef, ent_coef;
};

// End of an MFMA chain: within the chain's region the LDS reads (the NEXT chain's operands) go first, then the 16 MFMAs --
// left alone the scheduler sinks each ds_read_b128 to just before its first use, and the wave then stalls on an LDS round
// trip every four MFMAs; across regions nothing moves (operand loads hoisted from later chains would spill).
#ifndef R6U_NO_CHAIN_FENCE
#define R6U_CHAIN_FENCE()                                 \
    do {                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0); \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
#else
#define R6U_CHAIN_FENCE() do { } while (0)
#endif
#define R6U_OPAQUE(p) asm volatile("" : "+v"(p))
#ifdef R6U_STAMPS
// phase stamps of wave 0's last tile: raw 100 MHz ticks into kStampSlots extra floats of LDS behind the staging areas, copied out
// by workgroup 0 after the tile loop -- one s_memrealtime + ds_write each, cheap enough not to disturb the loop's registers
#define R6U_MARK(k) do { if (wave == 0) lds[Im::kTotal + (k)] = __uint_as_float((uint32_t)wall_clock64()); } while (0)
#else
#define R6U_MARK(k) do { } while (0)
#endif

template <int NETS, int H1, int H2, int OD>
__global__ __launch_bounds__(kBlock, 1) void ppo_grad_kernel(const R6PolicyNet net, const Args a, float* __restrict__ partial) {
#if R6U_DW2_DIRECT
    // MODE.FP16_OVFL: a conversion to f16 that overflows gives +-65504, not infinity.  The only values that can get there are delta2^T
    // entries of a tile 2^(15 - 13 + R6U_DW2_BASE_SHIFT) times the wave's first one and more, under the launch's factor: they saturate,
    // finite, at no instruction (a v_med3 in front of the conversion would undo its fusion with the scaling multiply: +100 per tile)
#ifndef R6U_NO_FP16_OVFL   // (timing experiments)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
#endif
#endif
    using Im = Image<NETS, H1, H2>;
    constexpr int B1 = H1 / 32, B2 = H2 / 32;
    [[maybe_unused]] constexpr int kP1 = Im::kP1;   // the fp32 forward's image pitch (R6U_F16_FWD=0)
    constexpr int kNetBlocks = B2 * B1 + B1;   // accumulator blocks of one net: dW2 [mb][ib], then dW1 [ib]
    static_assert(B2 == 2 && NETS * kNetBlocks == kNumBlocks && 5 * H2 == kFeatNet, "sized for a shared trunk [128, 64] or two MLPs (64, 64)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#ifdef R6U_STAMPS
    const uint64_t ts0 = wall_clock64();
#endif
    // ---- the first tiles' rows are asked for before anything else: their round trips run under the staging's ---------------
    const int64_t n_tiles = (a.n + 31) >> 5;
    // The gathers of a tile (its 32 rows of obs / action / old_logp / adv / ret, found through the index) are two dependent
    // memory round trips, and nobody hides them at one wave per SIMD: 1 us of a 21 us tile while the rollout buffer sits in
    // L2, several when it is 2^24 rows in HBM.  So they run one tile ahead: the next tile's row number is requested at the
    // top of a tile, its data after the loss (those registers are free during the backward pass), and both are consumed
    // in the next trip.  Row -1 = no sample (padding lane or no further tile).
    const int64_t t_stride = (int64_t)gridDim.x * kWaves;
    auto fetch_row = [&](int64_t tt) __attribute__((always_inline)) -> int {
        const int64_t sidx = tt * 32 + col;
        return (tt < n_tiles && sidx < a.n) ? (int)(a.index ? a.index[sidx] : sidx) : -1;
    };
    auto fetch_data = [&](int r, float (&x)[8], float (&sv)[6]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = R6U_F16_FWD ? s + 8 * h : 2 * s + h;   // the K layout of the layer-1 MFMA: 8 consecutive features per lane half (f16, K = 16) or interleaved (f32, K = 2 per step)
            x[s] = 0.0f;
            if (r >= 0) {
                if (k < OD) x[s] = a.obs[(int64_t)r * OD + k];
                else if (k == OD) x[s] = 1.0f;   // the bias column
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) sv[q] = 0.0f;
        if (r >= 0) {
            sv[0] = a.action[3 * (int64_t)r + 0];
            sv[1] = a.action[3 * (int64_t)r + 1];
            sv[2] = a.action[3 * (int64_t)r + 2];
            sv[3] = a.old_logp[r];
            sv[4] = a.adv[r];
            sv[5] = a.ret[r];
        }
    };
    const int64_t t_first = (int64_t)blockIdx.x * kWaves + wave;
    int row_cur = fetch_row(t_first), row_next = fetch_row(t_first + t_stride);
    float xk[8], sv[6];   // x: this lane's 8 features k = 2s + h of its sample (B operand of layer 1); sv: action[3], old_logp, adv, ret
    // ---- stage the weight images -------------------------------------------------------------------------------------
#if R6U_FAST_STAGING
    // Every workgroup builds the same images from the same ~40 KB of parameters, and at one wave per SIMD nobody hides a round
    // trip to L2 (about 1 us with 256 workgroups asking at once): so ALL of a thread's loads are issued first -- W2 as eight
    // 16-byte loads per thread, each of them four consecutive k of one row = one ds_write_b128 of the layer-2 image and four
    // lanes' halves of a delta-chain fragment -- and the LDS writes follow in one go.  (As four dependent loops this took
    // 7.2 us of a 55 us launch: scripts/ppo_grad_fixed_cost.py.)
    [[maybe_unused]] constexpr int kW2V = NETS * H2 * H1 / (4 * kBlock), kW1N = NETS * B1 * 64 * 8 / kBlock, kW3N = (Im::kW3 + kBlock - 1) / kBlock;
    static_assert(NETS * H2 * H1 % (4 * kBlock) == 0 && NETS * B1 * 64 * 8 % kBlock == 0 && Im::kB2 <= kBlock && H1 % 4 == 0, "whole rounds");
    constexpr int kFragN = NETS * B1 * B2 * 2 / kWaves;   // delta-chain fragments (chain, K-step) per wave
    static_assert(NETS * B1 * B2 * 2 % kWaves == 0, "whole rounds");
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 1
    uint64_t tsa = 0, tsb = 0, tsc = 0;
#endif
    auto put_frag = [&](int frag_float_off, const float (&v)[8]) __attribute__((always_inline)) {   // a lane's 8 halves of a fragment: hi copy, then lo copy
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (_Float16)v[j];
            lo[j] = (_Float16)(v[j] - (float)hi[j]);
        }
        h8* dst = reinterpret_cast<h8*>(lds) + frag_float_off / 4 + lane;
        dst[0] = hi;
        dst[Im::kFrag / 4] = lo;
    };
    float w2tv[kFragN][8];
    float w3v[kW3N], b2v = 0.0f;
#if R6U_F16_FWD
    // f16 forward: layer 2's fragments like the delta chain's, lane = (row m of W2, lane half), the 8 halves of a K-step = two runs
    // of four consecutive k of that row; layer 1 has one fragment per chain (K = 16 covers the observation and the bias column)
    static_assert(NETS * B1 == kWaves, "one layer-1 fragment per wave");
    float w2fv[kFragN][8], w1fv[8];
#else
    f32x4 w2v[kW2V];
    float w1v[kW1N];
#endif
    {
#if R6U_F16_FWD
#pragma unroll
        for (int i = 0; i < kFragN; ++i) {
            const int fi = wave + kWaves * i, chain = fi >> 1, st = fi & 1;   // chain (net * B2 + mb) * B1 + b
            const int b = chain % B1, mb = (chain / B1) % B2, ni = chain / (B2 * B1);
            const float* w = (ni ? net.vf_w2 : net.pi_w2) + (mb * 32 + col) * H1 + 32 * b;
#pragma unroll
            for (int j = 0; j < 8; ++j) w2fv[i][j] = w[rowmap(8 * st + j, h)];
        }
        {
            const int ni = wave / B1, blk = wave % B1, m = blk * 32 + col;
            const float* w = (ni ? net.vf_w1 : net.pi_w1) + m * OD;
            const float* bb = (ni ? net.vf_b1 : net.pi_b1) + m;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * h + j;
                const float v = *((k < OD) ? w + k : bb);
                w1fv[j] = (k <= OD) ? v : 0.0f;
            }
        }
#else
        const bool vec = ((reinterpret_cast<uintptr_t>(net.pi_w2) | (NETS == 2 ? reinterpret_cast<uintptr_t>(net.vf_w2) : 0)) & 15) == 0;
#pragma unroll
        for (int i = 0; i < kW2V; ++i) {
            const int f = tid + kBlock * i, ni = f / (H2 * H1 / 4), loc = f % (H2 * H1 / 4);
            const float* w = (ni ? net.vf_w2 : net.pi_w2) + 4 * loc;
            if (vec) {
                w2v[i] = *reinterpret_cast<const f32x4*>(w);
            } else {   // parameters that are views at odd offsets of one flat buffer
                w2v[i][0] = w[0];
                w2v[i][1] = w[1];
                w2v[i][2] = w[2];
                w2v[i][3] = w[3];
            }
        }
#pragma unroll
        for (int i = 0; i < kW1N; ++i) {
            const int e = tid + kBlock * i;
            const int sidx = e & 7, l = (e >> 3) & 63, chain = e >> 9;   // chain = net * B1 + blk
            const int ni = chain / B1, blk = chain % B1;
            const int k = 2 * sidx + (l >> 5), m = blk * 32 + (l & 31);
            const float* w = ni ? net.vf_w1 : net.pi_w1;
            const float* b = ni ? net.vf_b1 : net.pi_b1;
            const float v = *((k < OD) ? w + m * OD + k : b + m);   // one unconditional load (a load under a lane mask makes the compiler wait for everything in flight)
            w1v[i] = (k <= OD) ? v : 0.0f;
        }
#endif
        // W2 once more, as the delta chain wants it: a wave builds whole fragments, lane = (column n of W2, lane half), its 8 halves
        // = 8 rows of that column -- eight loads that are contiguous across the lanes, and one 16-byte LDS write each for the hi
        // and the lo copy.  (Scattering the row-major float4s above into the fragments as 2-byte writes ran into 8- to 16-way
        // bank conflicts: 64 ds_write_b16 per thread cost more than the loads.)
#pragma unroll
        for (int i = 0; i < kFragN; ++i) {
            const int fi = wave + kWaves * i, chain = fi >> 1, st = fi & 1;   // chain (net * B1 + ib) * B2 + c
            const int c = chain % B2, ib = (chain / B2) % B1, ni = chain / (B2 * B1);
            const float* w = (ni ? net.vf_w2 : net.pi_w2) + ib * 32 + col;
#pragma unroll
            for (int j = 0; j < 8; ++j) w2tv[i][j] = w[(32 * c + rowmap(8 * st + j, h)) * H1];
        }
        if (tid < Im::kB2) {   // ((net * B2 + mb) * 2 + h) * 16 + r
            const int r = tid & 15, hh = (tid >> 4) & 1, rest = tid >> 5;
            const int ni = rest / B2, mb = rest % B2;
            b2v = (ni ? net.vf_b2 : net.pi_b2)[mb * 32 + rowmap(r, hh)];
        }
#pragma unroll
        for (int i = 0; i < kW3N; ++i) {   // (((net * 2 + h) * B2 + mb) * 16 + r) * 4 + o
            const int e = tid + kBlock * i;
            const int o = e & 3, r = (e >> 2) & 15, rest = e >> 6;
            const int c = rest % B2, hh = (rest / B2) & 1, ni = rest / (2 * B2);
            const int k = 32 * c + rowmap(r, hh);
            static_assert(Im::kW3 % kBlock == 0, "whole rounds");
            const bool act_head = o < 3;
            const float v = *(act_head ? net.pi_w3 + o * H2 + k : net.vf_w3 + k);
            w3v[i] = (NETS == 1 || (ni == 0) == act_head) ? v : 0.0f;   // two MLPs: net 0 feeds the action head only, net 1 the value head only
        }
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 1
        __builtin_amdgcn_sched_barrier(0);
        tsa = wall_clock64();
        __builtin_amdgcn_sched_barrier(0);
#endif
        fetch_data(row_cur, xk, sv);   // the first tile's samples: requested behind the parameters, consumed after the barrier
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 1
        __builtin_amdgcn_sched_barrier(0);
        tsb = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tsc = wall_clock64();
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (tid < Im::kB2) lds[Im::oB2 + tid] = b2v;
#pragma unroll
        for (int i = 0; i < kW3N; ++i) {
            lds[Im::oW3 + tid + kBlock * i] = w3v[i];
        }
#if R6U_F16_FWD
        put_frag(Im::oW1 + wave * 2 * Im::kFrag, w1fv);
#pragma unroll
        for (int i = 0; i < kFragN; ++i) put_frag(Im::oW2 + (wave + kWaves * i) * 2 * Im::kFrag, w2fv[i]);
#else
#pragma unroll
        for (int i = 0; i < kW1N; ++i) {
            const int e = tid + kBlock * i;
            lds[Im::oW1 + (e >> 3) * kP1 + (e & 7)] = w1v[i];   // (chain * 64 + l) * kP1 + s
        }
#pragma unroll
        for (int i = 0; i < kW2V; ++i) {
            const int f = tid + kBlock * i, ni = f / (H2 * H1 / 4), loc = f % (H2 * H1 / 4);
            const int m = loc / (H1 / 4), k0 = 4 * (loc % (H1 / 4));          // W2[m][k0 .. k0 + 3]
            // layer-2 image, chain (net * B2 + mb) * B1 + b: lane (m % 32) + 32 h, registers r0 .. r0 + 3 with rowmap(r0 + i, h) = k0 % 32 + i
            const int chain = (ni * B2 + (m >> 5)) * B1 + (k0 >> 5);
            const int l = (m & 31) + 32 * ((k0 >> 2) & 1), r0 = 4 * ((k0 >> 3) & 3);
            *reinterpret_cast<f32x4*>(lds + Im::oW2 + (chain * 64 + l) * kP2 + r0) = w2v[i];
        }
#endif
#pragma unroll
        for (int i = 0; i < kFragN; ++i) put_frag(Im::oW2T + (wave + kWaves * i) * 2 * Im::kFrag, w2tv[i]);   // fragment (chain * 2 + step)
    }
#else
#if R6U_F16_ANY
    _Float16* ldsh = reinterpret_cast<_Float16*>(lds);
    auto put_split = [&](int frag_float_off, int e, float v) __attribute__((always_inline)) {   // e = lane * 8 + elem within the fragment
        const _Float16 hi = (_Float16)v;
        ldsh[2 * frag_float_off + e] = hi;
        ldsh[2 * (frag_float_off + Im::kFrag) + e] = (_Float16)(v - (float)hi);
    };
#endif
#if R6U_F16_FWD
#pragma unroll 8
    for (int e = tid; e < NETS * B1 * 64 * 8; e += kBlock) {
        const int j = e & 7, l = (e >> 3) & 63, chain = e >> 9;   // chain = net * B1 + blk
        const int ni = chain / B1, blk = chain % B1;
        const int k = 8 * (l >> 5) + j, m = blk * 32 + (l & 31);
        const float* w = ni ? net.vf_w1 : net.pi_w1;
        const float* b = ni ? net.vf_b1 : net.pi_b1;
        put_split(Im::oW1 + chain * 2 * Im::kFrag, e & 511, (k < OD) ? w[m * OD + k] : ((k == OD) ? b[m] : 0.0f));
    }
#else
#pragma unroll 8
    for (int e = tid; e < NETS * B1 * 64 * 8; e += kBlock) {
        const int s = e & 7, l = (e >> 3) & 63, chain = e >> 9;   // chain = net * B1 + blk
        const int ni = chain / B1, blk = chain % B1;
        const int k = 2 * s + (l >> 5), m = blk * 32 + (l & 31);
        const float* w = ni ? net.vf_w1 : net.pi_w1;
        const float* b = ni ? net.vf_b1 : net.pi_b1;
        lds[Im::oW1 + (chain * 64 + l) * kP1 + s] = (k < OD) ? w[m * OD + k] : ((k == OD) ? b[m] : 0.0f);
    }
#endif
#pragma unroll 8   // independent gathers from L2: eight in flight per thread instead of one round trip per element
    for (int e = tid; e < NETS * B2 * B1 * 64 * 16; e += kBlock) {
        // f32 layout: e = (chain * 64 + l) * 16 + r;  f16 layout: e = ((chain * 2 + step) * 64 + l) * 8 + j, r = 8 step + j
        const int rest = e >> 10;                      // chain
        const int ni = rest / (B2 * B1), loc = rest % (B2 * B1);
        const float* w = ni ? net.vf_w2 : net.pi_w2;
        const int r32 = e & 15, l32 = (e >> 4) & 63;
        const int j16 = e & 7, l16 = (e >> 3) & 63, st16 = (e >> 9) & 1, r16 = 8 * st16 + j16;
        {   // w2: chain (net * B2 + mb) * B1 + b
            const int b = loc % B1, mb = loc / B1;
#if R6U_F16_FWD
            put_split(Im::oW2 + (rest * 2 + st16) * 2 * Im::kFrag, e & 511, w[(mb * 32 + (l16 & 31)) * H1 + 32 * b + rowmap(r16, l16 >> 5)]);
#else
            lds[Im::oW2 + (rest * 64 + l32) * kP2 + r32] = w[(mb * 32 + (l32 & 31)) * H1 + 32 * b + rowmap(r32, l32 >> 5)];
#endif
        }
        {   // w2t: chain (net * B1 + ib) * B2 + c
            const int c = loc % B2, ib = loc / B2;
#if R6U_F16_BWD
            put_split(Im::oW2T + (rest * 2 + st16) * 2 * Im::kFrag, e & 511, w[(32 * c + rowmap(r16, l16 >> 5)) * H1 + ib * 32 + (l16 & 31)]);
#else
            lds[Im::oW2T + (rest * 64 + l32) * kP2 + r32] = w[(32 * c + rowmap(r32, l32 >> 5)) * H1 + ib * 32 + (l32 & 31)];
#endif
        }
    }
    for (int e = tid; e < Im::kB2; e += kBlock) {   // ((net * B2 + mb) * 2 + h) * 16 + r
        const int r = e & 15, hh = (e >> 4) & 1, rest = e >> 5;
        const int ni = rest / B2, mb = rest % B2;
        lds[Im::oB2 + e] = (ni ? net.vf_b2 : net.pi_b2)[mb * 32 + rowmap(r, hh)];
    }
    for (int e = tid; e < Im::kW3; e += kBlock) {   // (((net * 2 + h) * B2 + mb) * 16 + r) * 4 + o
        const int o = e & 3, r = (e >> 2) & 15, rest = e >> 6;
        const int c = rest % B2, hh = (rest / B2) & 1, ni = rest / (2 * B2);
        const int k = 32 * c + rowmap(r, hh);
        float v;
        if (NETS == 1) v = (o < 3) ? net.pi_w3[o * H2 + k] : net.vf_w3[k];
        else if (ni == 0) v = (o < 3) ? net.pi_w3[o * H2 + k] : 0.0f;
        else v = (o == 3) ? net.vf_w3[k] : 0.0f;
        lds[Im::oW3 + e] = v;
    }
#endif
    __syncthreads();

#ifdef R6U_STAMPS
    const uint64_t ts1 = wall_clock64();
#endif
    const int pcol = (col & 1) * 16 + (col >> 1);        // a sample's place in a transposed row: even samples first, then the odd ones
    const int hoff = h * 16;                             // the 16 places an MFMA chain of this lane half reads (sample 2s + h <-> h*16 + s)
    // One base register per access pattern, re-made opaque in every tile (R6U_OPAQUE): every LDS address below is then
    // base + an immediate.  Left to itself the compiler forms the 32 + 32 + ... addresses once, outside the tile loop, and
    // parks them in scratch -- a scratch reload in front of every ds_read, each an exposed round trip at one wave per SIMD
    // (measured: 8.5 us of a 36 us tile in the heads alone).
    // (The opaque values are element offsets into `lds`, not pointers: a pointer that went through an asm statement has
    // lost its address space and every access through it becomes a flat_load / flat_store.)
    const int stage0 = Im::kImage + wave * Im::kStage;
    // Offsets of 16-byte reads are kept in units of four floats (q_*), so that the compiler still sees the alignment behind
    // the asm statement and emits ds_read_b128 (it falls back to pairs of ds_read2_b32 otherwise).
    static_assert(Im::oW3 % 4 == 0 && Im::kImage % 4 == 0 && Im::kStage % 4 == 0 && kPitch % 4 == 0 && Im::sG % 4 == 0, "16-byte aligned rows");
    int q_w3 = (Im::oW3 + h * (B2 * 64)) >> 2;
    static_assert(Im::oB2 % 4 == 0, "16-byte aligned rows");
    int q_b2 = (Im::oB2 + h * 16) >> 2;
    int o_tw = stage0 + 4 * h * kPitch + pcol;           // writes of transposed tiles: element (row(r, h), sample)
    // delta1 and x feed the 16x16x4 product: there a lane group g = l / 16 contracts the samples g, g + 4, ..., g + 28, so their
    // rows keep sample s at place (s % 4) * 8 + s / 4 (8 consecutive floats per group = 2 ds_read_b128)
    int o_tw4 = stage0 + 4 * h * kPitch + (col & 3) * 8 + (col >> 2);
    int q_tr4 = (stage0 + (lane & 15) * kPitch + (lane >> 4) * 8) >> 2;   // reads: row l % 16 (+ 16 for the second block), group l / 16
    int q_tr = (stage0 + col * kPitch + hoff) >> 2;      // reads of row `col`, this lane half's 16 samples
    // wave-uniform scalars live in SGPRs (readfirstlane: the compiler keeps a computed float in a VGPR otherwise)
    auto uni = [](float v) __attribute__((always_inline)) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float b3_0 = uni(net.pi_b3[0]), b3_1 = uni(net.pi_b3[1]), b3_2 = uni(net.pi_b3[2]), b3_v = uni(net.vf_b3[0]);
    const float ls0 = uni(net.log_std[0]), ls1 = uni(net.log_std[1]), ls2 = uni(net.log_std[2]);
    const float iv0 = uni(__expf(-2.0f * ls0)), iv1 = uni(__expf(-2.0f * ls1)), iv2 = uni(__expf(-2.0f * ls2));
    const float ls_sum = uni(ls0 + ls1 + ls2 + 2.756815599614018f);
    const float a_mean = uni(a.adv_norm ? a.adv_norm[0] : 0.0f), a_istd = uni(a.adv_norm ? a.adv_norm[1] : 1.0f);
    const float inv_n = uni(1.0f / (float)a.n);
    const float vf2n = uni(a.vf_coef * 2.0f * inv_n);
    // The 11 running sums of a lane (db3[4], dlog_std[3], pg, v, clip_frac, kl) live in LDS, in the padding of the w2 image
    // (pitch kP2 = 20, 16 used): lane l of wave w owns float 16 + w of its row in chain k.  In registers they were the
    // first thing the allocator parked in scratch (11 reload-add-spill round trips per tile).
#if R6U_F16_ANY
    constexpr int kStatStride = kWaves * 32;                 // f16 images have no padding: the sums have an area of their own, one slot per sample-owning lane
    int o_stat = Im::oStat + wave * 32 + col;
#else
    constexpr int kStatStride = 64 * kP2;
    static_assert(kP2 - 16 >= kWaves && 2 * NETS * B2 * B1 >= 11 && Im::oW2T == Im::oW2 + Im::kW2, "stat slots fit into the padding of the w2 and w2t images (contiguous)");
    int o_stat = Im::oW2 + lane * kP2 + 16 + wave;
#endif
#pragma unroll
    for (int k = 0; k < 11; ++k) lds[o_stat + k * kStatStride] = 0.0f;

    // ---- accumulators of the launch -----------------------------------------------------------------------------------
    f32x16 acc[NETS * B2 * B1];                              // dW2 [net][mb][ib]
    f32x4 acc1[NETS * B1][2];                                // dW1 [net][ib]: two 16 x 16 blocks (rows 0..15, 16..31 of the 32-feature block)
    auto iW2 = [](int ni, int mb, int ib) __attribute__((always_inline)) { return (ni * B2 + mb) * B1 + ib; };
#if R6U_DW2_DIRECT
    constexpr int kNoBase = -1000;
    int e_base[NETS];                                        // the launch-long factor of a net's dW2 sums is 2^(13 - e_base); kNoBase: no tile yet
#pragma unroll
    for (int ni = 0; ni < NETS; ++ni) e_base[ni] = kNoBase;
#endif
    float f_w3[NETS][4], f_b2[NETS];                         // lane = layer-2 feature: dW3^T[lane][0..3], db2[lane]
#pragma unroll
    for (int ni = 0; ni < NETS; ++ni) {
        f_b2[ni] = 0.0f;
#pragma unroll
        for (int o = 0; o < 4; ++o) f_w3[ni][o] = 0.0f;
    }
    int q_fr = (stage0 + lane * kPitch) >> 2;                // row `lane` of the transposed a2 / delta2 tile
    int q_g = (stage0 + Im::sG) >> 2;                        // g's four rows (the same address in every lane: LDS broadcast)
#pragma unroll
    for (int k = 0; k < NETS * B2 * B1; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < NETS * B1; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[k][0][r] = acc1[k][1][r] = 0.0f;
    }

#if !R6U_FAST_STAGING
    fetch_data(row_cur, xk, sv);
#endif
#pragma unroll 1
    for (int64_t t = t_first; t < n_tiles; t += t_stride) {
        const bool live = row_cur >= 0;
        const int row_after = fetch_row(t + 2 * t_stride);   // arrives during this tile; becomes row_next at the bottom

        R6U_OPAQUE(q_w3);
        R6U_OPAQUE(q_b2);
        R6U_OPAQUE(o_tw);
        R6U_OPAQUE(o_tw4);
        R6U_OPAQUE(q_tr4);
        R6U_OPAQUE(q_tr);
        R6U_OPAQUE(o_stat);
        R6U_OPAQUE(q_fr);
        R6U_OPAQUE(q_g);
        const int o_fr = q_fr << 2, o_g = q_g << 2;
        const float* w3_base = lds + (q_w3 << 2);
        const float* b2_base = lds + (q_b2 << 2);
        float* tw_base = lds + o_tw;
        float* tw4_base = lds + o_tw4;
        const float* tr4_base = lds + (q_tr4 << 2);
        const float* tr_base = lds + (q_tr << 2);
        float* stat_base = lds + o_stat;
        wave_sync();   // the previous tile's readers of the staging area are done
#pragma unroll
        for (int s = 0; s < 8; ++s) tw4_base[Im::sX + (R6U_F16_FWD ? s + 4 * h : 2 * s - 3 * h) * kPitch] = xk[s];   // row k of this lane's feature (the base sits at row 4h)
        const float act0 = sv[0], act1 = sv[1], act2 = sv[2], olp = sv[3], advv = sv[4], retv = sv[5];
#ifdef R6U_STAMPS
        R6U_MARK(0);
#endif
        // ---- forward: one net after the other (the two MLPs share nothing but x) ------------------------------------------
        f32x16 a1[NETS][B1], a2[NETS][B2];
#if R6U_F16_ANY
        const h8* frag = reinterpret_cast<const h8*>(lds);          // fragment copies are 64 lanes x 16 bytes: index = float offset / 4 + lane
#endif
#if R6U_F16_FWD
        const Split xs = split8(xk);
#pragma unroll
        for (int ni = 0; ni < NETS; ++ni) {
#if R6U_SCHED & 32
            {   // layer 1 as a pipeline as well: all fragments requested at once, block b + 1's three MFMAs between the halves of block b's tanh
                h8 w1h[B1], w1l[B1];
#pragma unroll
                for (int blk = 0; blk < B1; ++blk) {
                    const int f = (Im::oW1 + (ni * B1 + blk) * 2 * Im::kFrag) / 4 + lane;
                    w1h[blk] = frag[f];
                    w1l[blk] = frag[f + Im::kFrag / 4];
                }
                f32x16 zc, zn;
#pragma unroll
                for (int r = 0; r < 16; ++r) zc[r] = 0.0f;
                zc = mfma3(w1h[0], w1l[0], xs, zc);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int blk = 0; blk < B1; ++blk) {
                    if (blk + 1 < B1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) zn[r] = 0.0f;
                        zn = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[blk + 1], xs.hi, zn, 0, 0, 0);
                        tanh_half(zc, a1[ni][blk], 0);
                        __builtin_amdgcn_sched_barrier(0);
                        zn = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[blk + 1], xs.lo, zn, 0, 0, 0);
                        tanh_half(zc, a1[ni][blk], 8);
                        __builtin_amdgcn_sched_barrier(0);
                        zn = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[blk + 1], xs.hi, zn, 0, 0, 0);
                        zc = zn;
                    } else {
                        tanh_block(zc, a1[ni][blk]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int blk = 0; blk < B1; ++blk) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                const int f = (Im::oW1 + (ni * B1 + blk) * 2 * Im::kFrag) / 4 + lane;
                z = R6U_FWD_MFMA(frag[f], frag[f + Im::kFrag / 4], xs, z);
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[ni][blk][r] = tanh_fast(z[r]);
#if !(R6U_SCHED & 1)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
#endif
            f32x16 z2[B2];
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) {
                float zi[16];
                ldv<16>(b2_base + (ni * B2 + mb) * 32, zi);
#pragma unroll
                for (int r = 0; r < 16; ++r) z2[mb][r] = zi[r];
            }
#if R6U_SCHED & 8
            {   // software pipeline over the 2 B1 K-steps: the next step's fragments are requested and its activations split while this
                // step's six MFMAs run -- one MFMA, five vector instructions, in that order (a chain of MFMAs issued back to back
                // holds the wave's issue for its whole duration, and the splits then run with the matrix pipe idle)
                constexpr int kG = 2 * B1;
                static_assert(R6U_F16_FWD, "f16 forward");
                Split bs[2];
                h8 wh[2][B2], wl[2][B2];
                auto splitg = [&](int g) __attribute__((always_inline)) -> Split {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a1[ni][g >> 1][(g & 1) * 8 + j];
                    return split8(v);
                };
                auto loadg = [&](int g, h8 (&hh)[B2], h8 (&ll)[B2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) {
                        const int f = (Im::oW2 + ((((ni * B2 + mb) * B1 + (g >> 1)) * 2 + (g & 1)) * 2) * Im::kFrag) / 4 + lane;
                        hh[mb] = frag[f];
                        ll[mb] = frag[f + Im::kFrag / 4];
                    }
                };
                loadg(0, wh[0], wl[0]);
                bs[0] = splitg(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < kG; ++g) {
                    if (g + 1 < kG) {
                        loadg(g + 1, wh[(g + 1) & 1], wl[(g + 1) & 1]);
                        bs[(g + 1) & 1] = splitg(g + 1);
                    }
                    // the three products of mfma3, the two blocks alternating: the same sums in the same order per accumulator
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) z2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[g & 1][mb], bs[g & 1].hi, z2[mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) z2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g & 1][mb], bs[g & 1].lo, z2[mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) z2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g & 1][mb], bs[g & 1].hi, z2[mb], 0, 0, 0);
                    if (g + 1 < kG) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2 * B2, 0);
#pragma unroll
                        for (int i = 0; i < 3 * B2; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int b = 0; b < B1; ++b) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a1[ni][b][sk * 8 + j];
                    const Split bs = split8(v);
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) {
                        const int f = (Im::oW2 + ((((ni * B2 + mb) * B1 + b) * 2 + sk) * 2) * Im::kFrag) / 4 + lane;
                        z2[mb] = R6U_FWD_MFMA(frag[f], frag[f + Im::kFrag / 4], bs, z2[mb]);
                    }
#if !(R6U_SCHED & 2)
                    asm volatile("" ::: "memory");   // keeps the scheduler from hoisting every group's LDS reads up front
#endif
                }
#if !(R6U_SCHED & 4)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
#endif
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) tanh_block(z2[mb], a2[ni][mb]);
        }
#else
#pragma unroll
        for (int ni = 0; ni < NETS; ++ni) {
#pragma unroll
            for (int blk = 0; blk < B1; ++blk) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                float wa[8];
                ldv<8>(&lds[Im::oW1 + ((ni * B1 + blk) * 64 + lane) * kP1], wa);
#pragma unroll
                for (int s = 0; s < 8; ++s) z = mfma(wa[s], xk[s], z);
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[ni][blk][r] = tanh_fast(z[r]);
                R6U_CHAIN_FENCE();   // one chain at a time: operand loads hoisted from later chains would spill, and with
                                     // one wave per SIMD a scratch access is an exposed memory round trip
            }
            {   // chains c = mb * B1 + b; the next chain's 16 weights are fetched while this chain runs on the matrix core
                constexpr int kC = B2 * B1;
                float wa[2][16];
                ldv<16>(&lds[Im::oW2 + (ni * kC * 64 + lane) * kP2], wa[0]);
                f32x16 z;
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    const int mb = c / B1, b = c % B1;
                    if (b == 0) {
                        float zi[16];
                        ldv<16>(b2_base + (ni * B2 + mb) * 32, zi);
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = zi[r];
                    }
                    if (c + 1 < kC) ldv<16>(&lds[Im::oW2 + ((ni * kC + c + 1) * 64 + lane) * kP2], wa[(c + 1) & 1]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) z = mfma(wa[c & 1][r], a1[ni][b][r], z);
                    if (b == B1 - 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) a2[ni][mb][r] = tanh_fast(z[r]);
                    }
                    R6U_CHAIN_FENCE();
                }
            }
        }
#endif
        R6U_MARK(1);   // forward
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ni = 0; ni < NETS; ++ni) {
            // LDS operands in batches: all 16 rows of a block are requested before the first is used, the next block's while this
            // one's FMAs issue -- read-use-read-use exposed an LDS round trip per row at one wave per SIMD (two thirds of the
            // kernel's s_waitcnt sat within two instructions of their ds_read)
            constexpr int kCh = 2 * B2;               // chunks of 8 rows (row = 16 mb + r)
            f32x4 w3r[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w3r[0][j] = *reinterpret_cast<const f32x4*>(w3_base + ni * (2 * B2 * 64) + j * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < kCh; ++c) {
                if (c + 1 < kCh) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) w3r[(c + 1) & 1][j] = *reinterpret_cast<const f32x4*>(w3_base + ni * (2 * B2 * 64) + ((c + 1) * 8 + j) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float av = a2[ni][c >> 1][(c & 1) * 8 + j];
                    part[0] = fmaf(w3r[c & 1][j][0], av, part[0]);
                    part[1] = fmaf(w3r[c & 1][j][1], av, part[1]);
                    part[2] = fmaf(w3r[c & 1][j][2], av, part[2]);
                    part[3] = fmaf(w3r[c & 1][j][3], av, part[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float m0 = part[0] + __shfl_xor(part[0], 32) + b3_0;
        const float m1 = part[1] + __shfl_xor(part[1], 32) + b3_1;
        const float m2 = part[2] + __shfl_xor(part[2], 32) + b3_2;
        const float vv = part[3] + __shfl_xor(part[3], 32) + b3_v;

        // ---- loss and its gradient w.r.t. (mean, value): r6_ppo_loss_grad's formulas, per sample --------------------------
        float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (live) {
            const float d0 = act0 - m0, d1 = act1 - m1, d2 = act2 - m2;
            const float q0 = d0 * d0 * iv0, q1 = d1 * d1 * iv1, q2 = d2 * d2 * iv2;
            const float logp = -0.5f * (q0 + q1 + q2) - ls_sum;
            const float dl = logp - olp;
            const float ratio = __expf(dl);
            const float A = (advv - a_mean) * a_istd;
            const float s1 = A * ratio, s2 = A * fminf(fmaxf(ratio, 1.0f - a.clip), 1.0f + a.clip);
            const bool active = !((A > 0.0f && ratio > 1.0f + a.clip) || (A < 0.0f && ratio < 1.0f - a.clip));
            const float g_logp = active ? -s1 * inv_n : 0.0f;
            g[0] = g_logp * d0 * iv0;
            g[1] = g_logp * d1 * iv1;
            g[2] = g_logp * d2 * iv2;
            const float ve = vv - retv;
            g[3] = vf2n * ve;
            if (h == 0) {   // each sample is counted once (both lane halves hold it)
                const float add[11] = {g[0], g[1], g[2], g[3], g_logp * (q0 - 1.0f), g_logp * (q1 - 1.0f), g_logp * (q2 - 1.0f),
                                       -fminf(s1, s2), ve * ve, (fabsf(ratio - 1.0f) > a.clip) ? 1.0f : 0.0f, (ratio - 1.0f) - dl};
                float cur[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) cur[k] = stat_base[k * kStatStride];
#pragma unroll
                for (int k = 0; k < 11; ++k) stat_base[k * kStatStride] = cur[k] + add[k];
            }
        }

        R6U_MARK(2);   // heads + loss
        fetch_data(row_next, xk, sv);   // the next tile's rows: in flight during the backward pass
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) tw_base[Im::sG + o * kPitch] = g[o];
        }
        // ---- backward, one net after the other --------------------------------------------------------------------------------
#pragma unroll
        for (int ni = 0; ni < NETS; ++ni) {
            if (ni > 0) wave_sync();   // the previous net's readers of the tile buffers are done
            // dW3^T[f][o] += sum_s a2[f][s] g[o][s]: a2 transposed through LDS; on the VALU, lane = feature (H2 = 64 rows of
            // the tile, one per lane): 128 FMAs -- as an MFMA product with 4 useful columns of 32 this took 32 matrix
            // instructions (2048+ cycles) and two accumulator blocks
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tw_base[Im::sT2 + (mb * 32 + rowmap(r, 0)) * kPitch] = a2[ni][mb][r];
            }
            wave_sync();
            if (!(R6U_ABLATE & 1)) {   // this lane's row of a2 (8 reads) and g's four rows (32 broadcast reads, two q at a time)
                f32x4 av[8], gv[2][2][4];
#pragma unroll
                for (int q = 0; q < 8; ++q) av[q] = *reinterpret_cast<const f32x4*>(lds + o_fr + Im::sT2 + 4 * q);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) gv[0][q][o] = *reinterpret_cast<const f32x4*>(lds + o_g + o * kPitch + 4 * q);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c + 1 < 4) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) gv[(c + 1) & 1][q][o] = *reinterpret_cast<const f32x4*>(lds + o_g + o * kPitch + 4 * (2 * (c + 1) + q));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 a4 = av[2 * c + q];
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const f32x4 g4 = gv[c & 1][q][o];
                            f_w3[ni][o] = fmaf(a4[0], g4[0], fmaf(a4[1], g4[1], fmaf(a4[2], g4[2], fmaf(a4[3], g4[3], f_w3[ni][o]))));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            R6U_MARK(3);   // dW3 sums (+ staging of a2)
            // ---- delta2 = (W3^T g) (1 - a2^2): this lane half's 16 rows per block ---------------------------------------------
            f32x16 d2[B2];
            {
                constexpr int kCh = 2 * B2;
                f32x4 w3r[2][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w3r[0][j] = *reinterpret_cast<const f32x4*>(w3_base + ni * (2 * B2 * 64) + j * 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < kCh; ++c) {
                    if (c + 1 < kCh) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) w3r[(c + 1) & 1][j] = *reinterpret_cast<const f32x4*>(w3_base + ni * (2 * B2 * 64) + ((c + 1) * 8 + j) * 4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int mb = c >> 1, r = (c & 1) * 8 + j;
                        const f32x4 w3 = w3r[c & 1][j];
                        const float up = fmaf(w3[0], g[0], fmaf(w3[1], g[1], fmaf(w3[2], g[2], w3[3] * g[3])));
                        d2[mb][r] = up * fmaf(-a2[ni][mb][r], a2[ni][mb][r], 1.0f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wave_sync();   // the dW3 sums' reads of the a2 tile are done: delta2 takes its place
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tw_base[Im::sT2 + (mb * 32 + rowmap(r, 0)) * kPitch] = d2[mb][r];
            }
            wave_sync();
            // db2[f] += sum_s delta2[f][s]: lane = feature again, a row sum
            {
                float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 dv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) dv[q] = *reinterpret_cast<const f32x4*>(lds + o_fr + Im::sT2 + 4 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < ((R6U_ABLATE & 4) ? 0 : 8); ++q) {
                    rs[0] += dv[q][0];
                    rs[1] += dv[q][1];
                    rs[2] += dv[q][2];
                    rs[3] += dv[q][3];
                }
                f_b2[ni] += (rs[0] + rs[1]) + (rs[2] + rs[3]);
            }
            R6U_MARK(4);   // delta2 + db2
            // ---- per 32-feature block of layer 1: dW2 (needs a1), delta1 (needs a1), dW1 (needs delta1 and x) -------------------
#if R6U_F16_BWD
            // delta2 as the f16 B operand of the delta1 chains: gradients are tiny (g carries 1 / n) and of any magnitude, so the
            // tile is scaled by a power of two first -- exact -- such that its largest entry sits in [2^13, 2^14): far from f16's
            // 65504, and every entry down to 2^-16 of the largest one still splits into a normal hi and a normal-or-exact lo
            Split d2s[B2][2];
            float d2_unscale, d2_scale;
            {
                float m = 0.0f;
#pragma unroll
                for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(d2[mb][r]));
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;     // floor(log2 m); -127 for zero / denormal
                if (e < -100) e = -100;                                      // nothing to scale: keep the factor finite
                const float scale = __uint_as_float((uint32_t)(127 + 13 - e) << 23);
                d2_scale = scale;
                d2_unscale = __uint_as_float((uint32_t)(127 - 13 + e) << 23);
#if R6U_DW2_DIRECT
                {
                    const int eu = __builtin_amdgcn_readfirstlane(e);   // (the same in every lane: the maximum went round the wave)
                    int eb = e_base[ni];
                    if (eb == kNoBase && eu > -100) {   // the wave's first tile with a non-zero delta2 sets the launch's factor: its largest entry to [2^(13 - shift), 2^(14 - shift))
                        eb = eu + R6U_DW2_BASE_SHIFT;
                        e_base[ni] = eb;
                    }
                    int fe = (eb == kNoBase) ? 0 : 13 - eb;
                    fe = fe < -126 ? -126 : (fe > 127 ? 127 : fe);
                    d2_scale = __uint_as_float((uint32_t)(127 + fe) << 23);   // dTs' factor from here on; d2s above has the tile's own
                }
#endif
#pragma unroll
                for (int c = 0; c < B2; ++c) {
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = d2[c][sk * 8 + j] * scale;
                        d2s[c][sk] = split8(v);
                    }
                }
            }
#endif
            float dT[2][16], bxc[8];   // delta2^T rows and x's row are the same for every block: fetched once per net and tile
            ldv<16>(tr_base + Im::sT2, dT[0]);
            ldv<16>(tr_base + Im::sT2 + 32 * kPitch, dT[1]);
            ldv<8>(tr4_base + Im::sX, bxc);   // x's row l % 16, samples g, g + 4, ...: B of the 16x16x4 product below
#if R6U_F16_DW2
            // dW2 on the f16 matrix pipe as well: a lane half's 16 samples of a row are two K-steps of 8 -- the operands the fp32
            // chains step through two at a time, split as they are.  delta2 goes in scaled like above, so a tile's products are
            // summed in a block of their own and join the launch's sums times the inverse factor (exact powers of two).
            Split dTs[B2][2];
#pragma unroll
            for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = dT[mb][8 * sk + j] * d2_scale;
                    dTs[mb][sk] = split8(v);
                }
            }
#endif
#if R6U_F16_DW2
            constexpr bool kPipeBlocks = (R6U_SCHED & 16) != 0 && NETS == 1;   // two MLPs: the pipelined form spills (11 registers) and loses
#else
            constexpr bool kPipeBlocks = false;
#endif
#if R6U_F16_DW2
            if constexpr (kPipeBlocks) {
            // The block loop as a software pipeline: matrix instructions and the vector work of the NEIGHBOURING stage alternate in
            // program order (a wave issues in order: MFMAs back to back hold its issue for their whole duration, and the vector
            // work then runs with the matrix pipe idle) --
            //   B  the delta1 chain (12 MFMAs)            under  the split of the a1 block (dW2's B operand)
            //   C  dW2's products (12 MFMAs, two chains)  under  delta1's scaling and tanh derivative
            //   E  dW1's products (16 MFMAs, two chains)  under  adding dW2's products to the launch's sums
            // Same operands, same order of sums per accumulator as the plain form (R6U_SCHED bit 16 off).
#pragma unroll
            for (int ib = 0; ib < B1; ++ib) {
                wave_sync();   // the block buffer's previous readers are done
#pragma unroll
                for (int r = 0; r < 16; ++r) tw_base[Im::sBlk + rowmap(r, 0) * kPitch] = a1[ni][ib][r];
                wave_sync();
                // ---- A: requests
                float ba[16];
                ldv<16>(tr_base + Im::sBlk, ba);
                h8 wh[B2][2], wl[B2][2];
#pragma unroll
                for (int c = 0; c < B2; ++c) {
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk) {
                        const int f = (Im::oW2T + ((((ni * B1 + ib) * B2 + c) * 2 + sk) * 2) * Im::kFrag) / 4 + lane;
                        wh[c][sk] = frag[f];
                        wl[c][sk] = frag[f + Im::kFrag / 4];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- B: delta1[i][s] = sum_f2 W2[f2][i] delta2[f2][s] (scaled), the a1 block split meanwhile: one MFMA of the chain,
                //      one pair of the block split, fenced -- the group-barrier form of this stage came out as runs of MFMAs followed
                //      by a run of 44 vector instructions
                f32x16 d1;
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
                Split bas[2];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int c = i / 6, sk = (i / 3) % 2, pr = i % 3;   // mfma3's order: lo * hi, hi * lo, hi * hi per (c, sk)
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? wl[c][sk] : wh[c][sk], pr == 1 ? d2s[c][sk].lo : d2s[c][sk].hi, d1, 0, 0, 0);
                    if (i < 8) {
                        const int ss = i / 4, j = 2 * (i % 4);
                        h2 hi2, lo2;
                        split2(ba[8 * ss + j], ba[8 * ss + j + 1], hi2, lo2);
                        bas[ss].hi[j] = hi2[0];
                        bas[ss].hi[j + 1] = hi2[1];
                        bas[ss].lo[j] = lo2[0];
                        bas[ss].lo[j + 1] = lo2[1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- C: dW2's products of this tile (its own block: delta2 went in scaled), delta1 finished meanwhile
#if R6U_DW2_DIRECT   // dW2's products go straight into the launch's sums (delta2^T went in under the launch's factor)
#define R6U_PART(mb) acc[iW2(ni, mb, ib)]
#else
                f32x16 part[B2];
#pragma unroll
                for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[mb][r] = 0.0f;
                }
#define R6U_PART(mb) part[mb]
#endif
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {   // mfma3's three products, the two blocks alternating
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) R6U_PART(mb) = __builtin_amdgcn_mfma_f32_32x32x16_f16(dTs[mb][sk].lo, bas[sk].hi, R6U_PART(mb), 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) R6U_PART(mb) = __builtin_amdgcn_mfma_f32_32x32x16_f16(dTs[mb][sk].hi, bas[sk].lo, R6U_PART(mb), 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) R6U_PART(mb) = __builtin_amdgcn_mfma_f32_32x32x16_f16(dTs[mb][sk].hi, bas[sk].hi, R6U_PART(mb), 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = (d1[r] * d2_unscale) * fmaf(-a1[ni][ib][r], a1[ni][ib][r], 1.0f);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#undef R6U_PART
                // ---- D: delta1 takes the a1 block's place (its readers -- the request of stage A -- are done)
                wave_sync();
#pragma unroll
                for (int r = 0; r < 16; ++r) tw4_base[Im::sBlk + rowmap(r, 0) * kPitch] = d1[r];
                wave_sync();
                float t0[8], t1[8];
                ldv<8>(tr4_base + Im::sBlk, t0);
                ldv<8>(tr4_base + Im::sBlk + 16 * kPitch, t1);
                __builtin_amdgcn_sched_barrier(0);
                // ---- E: dW1[i][k] += sum_s delta1[i][s] x[k][s] (16x16x4, two row blocks), dW2's products join the launch's sums meanwhile
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc1[ni * B1 + ib][0] = mfma16(t0[j], bxc[j], acc1[ni * B1 + ib][0]);
                    acc1[ni * B1 + ib][1] = mfma16(t1[j], bxc[j], acc1[ni * B1 + ib][1]);
                }
#if !R6U_DW2_DIRECT
#pragma unroll
                for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[iW2(ni, mb, ib)][r] = fmaf(part[mb][r], d2_unscale, acc[iW2(ni, mb, ib)][r]);
                }
#endif
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            } else
#endif
            {
#pragma unroll
            for (int ib = 0; ib < B1; ++ib) {
                wave_sync();   // the block buffer's previous readers are done
#pragma unroll
                for (int r = 0; r < 16; ++r) tw_base[Im::sBlk + rowmap(r, 0) * kPitch] = a1[ni][ib][r];
                wave_sync();
                // dW2[f2][i] += sum_s delta2[f2][s] a1[i][s]
                // Each chain's operands are fetched while the chain before it runs (static_assert: B2 == 2).
#if !R6U_F16_BWD
                float op[2][16];
#endif
                {
                    float ba[16];
                    ldv<16>(tr_base + Im::sBlk, ba);
#if !R6U_F16_BWD
                    ldv<16>(&lds[Im::oW2T + (((ni * B1 + ib) * B2 + 0) * 64 + lane) * kP2], op[0]);
#endif
#if R6U_F16_DW2
                    Split bas[2];
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = ba[8 * sk + j];
                        bas[sk] = split8(v);
                    }
#pragma unroll
                    for (int mb = 0; mb < B2; ++mb) {
#if R6U_DW2_DIRECT
#pragma unroll
                        for (int sk = 0; sk < 2; ++sk) acc[iW2(ni, mb, ib)] = mfma3(dTs[mb][sk].hi, dTs[mb][sk].lo, bas[sk], acc[iW2(ni, mb, ib)]);
#else
                        f32x16 part;
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[r] = 0.0f;
#pragma unroll
                        for (int sk = 0; sk < 2; ++sk) part = mfma3(dTs[mb][sk].hi, dTs[mb][sk].lo, bas[sk], part);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[iW2(ni, mb, ib)][r] = fmaf(part[r], d2_unscale, acc[iW2(ni, mb, ib)][r]);
#endif
                    }
#else
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[iW2(ni, 0, ib)] = mfma(dT[0][s], ba[s], acc[iW2(ni, 0, ib)]);
                    R6U_CHAIN_FENCE();
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[iW2(ni, 1, ib)] = mfma(dT[1][s], ba[s], acc[iW2(ni, 1, ib)]);
                    R6U_CHAIN_FENCE();
#endif
                }
                R6U_MARK(5 + 3 * ib);   // G2
                // delta1[i][s] = (sum_f2 W2[f2][i] delta2[f2][s]) (1 - a1[i][s]^2)
                f32x16 d1;
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
#if R6U_F16_BWD
                {   // the block's eight fragments are requested in one go (read - wait - three MFMAs, four times over, exposed a round trip each)
                    h8 wh[B2][2], wl[B2][2];
#pragma unroll
                    for (int c = 0; c < B2; ++c) {
#pragma unroll
                        for (int sk = 0; sk < 2; ++sk) {
                            const int f = (Im::oW2T + ((((ni * B1 + ib) * B2 + c) * 2 + sk) * 2) * Im::kFrag) / 4 + lane;
                            wh[c][sk] = frag[f];
                            wl[c][sk] = frag[f + Im::kFrag / 4];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < B2; ++c) {
#pragma unroll
                        for (int sk = 0; sk < 2; ++sk) d1 = mfma3(wh[c][sk], wl[c][sk], d2s[c][sk], d1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = (d1[r] * d2_unscale) * fmaf(-a1[ni][ib][r], a1[ni][ib][r], 1.0f);
#else
                ldv<16>(&lds[Im::oW2T + (((ni * B1 + ib) * B2 + 1) * 64 + lane) * kP2], op[1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) d1 = mfma(op[0][r], d2[0][r], d1);
                R6U_CHAIN_FENCE();
#pragma unroll
                for (int r = 0; r < 16; ++r) d1 = mfma(op[1][r], d2[1][r], d1);
                R6U_CHAIN_FENCE();
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] *= fmaf(-a1[ni][ib][r], a1[ni][ib][r], 1.0f);
#endif
                R6U_MARK(6 + 3 * ib);   // B2
                wave_sync();   // dW2's reads of the a1 block are done: delta1 takes its place
#pragma unroll
                for (int r = 0; r < 16; ++r) tw4_base[Im::sBlk + rowmap(r, 0) * kPitch] = d1[r];
                wave_sync();
                // dW1[i][k] += sum_s delta1[i][s] x[k][s]   (k == OD: db1, through x's column of ones): 16x16x4, two row blocks
                {
                    float t0[8], t1[8];
                    ldv<8>(tr4_base + Im::sBlk, t0);
                    ldv<8>(tr4_base + Im::sBlk + 16 * kPitch, t1);
#pragma unroll
                    for (int j = 0; j < ((R6U_ABLATE & 2) ? 0 : 8); ++j) {
                        acc1[ni * B1 + ib][0] = mfma16(t0[j], bxc[j], acc1[ni * B1 + ib][0]);
                        acc1[ni * B1 + ib][1] = mfma16(t1[j], bxc[j], acc1[ni * B1 + ib][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                R6U_MARK(7 + 3 * ib);   // G1
            }
            }
        }
        row_cur = row_next;
        row_next = row_after;
    }
#if R6U_DW2_DIRECT
    // the launch's factor leaves the dW2 sums (exact; a wave without a tile holds zeros)
#pragma unroll
    for (int ni = 0; ni < NETS; ++ni) {
        int ue = (e_base[ni] == kNoBase) ? 0 : e_base[ni] - 13;
        ue = ue > 127 ? 127 : ue;
        const float up = __uint_as_float((uint32_t)(127 + ue) << 23);
#pragma unroll
        for (int mb = 0; mb < B2; ++mb) {
#pragma unroll
            for (int ib = 0; ib < B1; ++ib) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[iW2(ni, mb, ib)][r] *= up;
            }
        }
    }
#endif

    // ---- workgroup reduction: scalars, per-feature sums, then the accumulator blocks four at a time through the staging areas ----
#ifdef R6U_STAMPS
    const uint64_t ts2 = wall_clock64();
#endif
    float sc[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) sc[k] = (R6U_F16_ANY && h != 0) ? 0.0f : lds[o_stat + k * kStatStride];   // (own area: the two halves of an env share a slot)
    __syncthreads();                               // every wave has left the tile loop: images and staging areas are free
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 2
    const uint64_t tsa = wall_clock64();
#endif
    // One exchange, one barrier.  Wave q finishes the quarter q of the accumulators: dW2 blocks 2q and 2q + 1 and dW1 block q,
    // 40 registers.  Every wave parks the three quarters it does not own in LDS -- [source wave][quarter][10][lane] float4, read
    // back by the same lane of the owner, so any layout is conflict-free -- next to its scalars and per-feature sums; after the
    // barrier the owner adds the four contributions in the fixed order (w0 + w1) + (w2 + w3), its own from registers, and
    // stores the blocks in canonical [m][n] order.  No atomics, and the sum does not depend on timing.  (Round 2 drained four
    // blocks at a time through the staging areas, twelve barriers: 7 us of a 55 us launch -- scripts/ppo_grad_fixed_cost.py.)
    constexpr int kXchWave = 3 * 10 * 64 * 4, kXch = kWaves * kXchWave;   // floats
    constexpr int kF = NETS * kFeatNet;
    static_assert(kWaves == 4 && NETS * B2 * B1 == 8 && NETS * B1 == 4, "four quarters of two dW2 blocks and one dW1 block each");
    static_assert(kXch + kWaves * kScalars + kWaves * kF <= Im::kTotal, "the exchange fits into the images + staging areas");
    float* reds = lds + kXch;                      // [kWaves][kScalars]
    float* fr = reds + kWaves * kScalars;          // [kWaves][net][5][64]
    float* out = partial + (int64_t)blockIdx.x * kPartial;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {       // level by level: the 11 exchanges of a level are independent (sum by sum they were 66 dependent LDS round trips)
        float o[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) o[k] = __shfl_xor(sc[k], off);
#pragma unroll
        for (int k = 0; k < 11; ++k) sc[k] += o[k];
    }
    if (lane < 11) {
        float v = sc[0];
#pragma unroll
        for (int k = 1; k < 11; ++k) v = (lane == k) ? sc[k] : v;
        reds[wave * kScalars + lane] = v;
    }
#pragma unroll
    for (int ni = 0; ni < NETS; ++ni) {
#pragma unroll
        for (int o = 0; o < 4; ++o) fr[wave * kF + ni * kFeatNet + o * 64 + lane] = f_w3[ni][o];
        fr[wave * kF + ni * kFeatNet + 4 * 64 + lane] = f_b2[ni];
    }
    auto quarter = [&](int q, int v4) __attribute__((always_inline)) -> f32x4 {   // float4 v4 of quarter q of this wave's accumulators
        if (v4 < 8) {
            const f32x16& b = acc[2 * q + (v4 >> 2)];
            const int r = 4 * (v4 & 3);
            return f32x4{b[r], b[r + 1], b[r + 2], b[r + 3]};
        }
        return acc1[q][v4 - 8];
    };
    f32x4* xch = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q != wave) {
            const int slot = (q < wave) ? q : q - 1;
#pragma unroll
            for (int v4 = 0; v4 < 10; ++v4) xch[((wave * 3 + slot) * 10 + v4) * 64 + lane] = quarter(q, v4);
        }
    }
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t tsb = wall_clock64();
#endif
    __syncthreads();
#if defined(R6U_STAMPS) && R6U_STAMP_SET == 2
    const uint64_t tsc = wall_clock64();
#endif
    if (tid < kScalars) out[kOffScal + tid] = (tid < 11) ? (reds[tid] + reds[kScalars + tid]) + (reds[2 * kScalars + tid] + reds[3 * kScalars + tid]) : 0.0f;
    for (int e = tid; e < kF; e += kBlock) out[kOffFeat + e] = (fr[e] + fr[kF + e]) + (fr[2 * kF + e] + fr[3 * kF + e]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q == wave) {
            f32x4 part[10][4];                     // all 30 reads first, in one go: read-add-store group by group was ten exposed LDS round trips
#pragma unroll
            for (int v4 = 0; v4 < 10; ++v4) {
#pragma unroll
                for (int src = 0; src < 4; ++src) {
                    if (src != q) part[v4][src] = xch[((src * 3 + ((q < src) ? q : q - 1)) * 10 + v4) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v4 = 0; v4 < 10; ++v4) {
                part[v4][q] = quarter(q, v4);
                const f32x4 tot = (part[v4][0] + part[v4][1]) + (part[v4][2] + part[v4][3]);
                if (v4 < 8) {   // dW2 block 2q + v4 / 4 in the partial set's net-major order (per net: dW2 [mb][ib], then dW1 [ib])
                    const int ai = 2 * q + (v4 >> 2), blk = (ai / (B2 * B1)) * kNetBlocks + ai % (B2 * B1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[blk * 1024 + rowmap(4 * (v4 & 3) + i, h) * 32 + col] = tot[i];
                } else {        // dW1's 16 x 16 blocks: the left half of a 32 x 32 slot (the right half is never read)
                    const int blk = (q / B1) * kNetBlocks + B2 * B1 + q % B1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[blk * 1024 + ((v4 - 8) * 16 + 4 * (lane >> 4) + i) * 32 + (lane & 15)] = tot[i];
                }
            }
        }
    }
#ifdef R6U_STAMPS
    if (tid == 0) {   // 100 MHz ticks: staging, tile loop of wave 0, reduction
#if R6U_STAMP_SET == 1     // staging: parameter loads issued | the first tile's loads issued | everything arrived | LDS writes + barrier
        out[kOffScal + 12] = (float)(tsa - ts0);
        out[kOffScal + 13] = (float)(tsb - tsa);
        out[kOffScal + 14] = (float)(tsc - tsb);
        out[kOffScal + 15] = (float)(ts1 - tsc);
#elif R6U_STAMP_SET == 2   // reduction: first barrier | shuffles + exchange writes | second barrier | sums + stores issued
        out[kOffScal + 12] = (float)(tsa - ts2);
        out[kOffScal + 13] = (float)(tsb - tsa);
        out[kOffScal + 14] = (float)(tsc - tsb);
        out[kOffScal + 15] = (float)(wall_clock64() - tsc);
#else
        out[kOffScal + 12] = (float)(ts1 - ts0);
        out[kOffScal + 13] = (float)(ts2 - ts1);
        out[kOffScal + 14] = (float)(wall_clock64() - ts2);
#endif
    }
    if (tid < kStampSlots && blockIdx.x == 0) {   // differences to the tile's first stamp, behind the last workgroup's partial set
        const uint32_t t0 = __float_as_uint(lds[Im::kTotal]), tk = __float_as_uint(lds[Im::kTotal + tid]);
        partial[(int64_t)gridDim.x * kPartial + tid] = (float)(tk - t0);
    }
#endif
}

// Sums the partial sets of all workgroups and scatters them into torch's parameter layouts ([out][in] row-major).
template <int NETS, int H1, int H2, int OD>
__global__ __launch_bounds__(1024) void ppo_grad_reduce_kernel(const float* __restrict__ partial, const int n_sets, const R6PolicyGrad gr,
                                                              const float* __restrict__ log_std, const float ent_coef, const float inv_n,
                                                              const float n_samples, float* __restrict__ stats) {
    constexpr int B1 = H1 / 32, B2 = H2 / 32;
    // 64 entries per workgroup (coalesced across lanes), the partial sets split over the 16 waves: 256 sets = two rounds of
    // 8 loads in flight per lane (with 4 waves it was eight dependent rounds: 8.7 us for 12.9 MB)
    constexpr int kW = 16;
    __shared__ float red[kW][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float v = 0.0f;
    // the right half of a dW1 slot (a 16-wide block in a 32-wide slot) is never written: a sixth of a partial set that is not read
    const bool unused = e < kNumBlocks * 1024 && ((e >> 10) % (B2 * B1 + B1)) >= B2 * B1 && (e & 31) >= 16;
    if (e < kPartial && !unused) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int s = w;
        for (; s + 7 * kW < n_sets; s += 8 * kW) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += partial[(int64_t)(s + kW * u) * kPartial + e];
        }
        for (; s < n_sets; s += kW) acc[0] += partial[(int64_t)s * kPartial + e];
        v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[w][threadIdx.x & 63] = v;
    __syncthreads();
    if (w != 0 || e >= kPartial) return;
    v = 0.0f;
#pragma unroll
    for (int k = 0; k < kW; k += 4) v += (red[k][threadIdx.x] + red[k + 1][threadIdx.x]) + (red[k + 2][threadIdx.x] + red[k + 3][threadIdx.x]);
    if (e < kNumBlocks * 1024) {
        constexpr int kNetBlocks = B2 * B1 + B1;
        const int blk = e >> 10, m = (e >> 5) & 31, nn = e & 31;
        const int ni = blk / kNetBlocks, loc = blk % kNetBlocks;   // net 1 (two MLPs): the value net's trunk
        if (loc < B2 * B1) {                       // dW2[mb*32 + m][ib*32 + n]
            const int mb = loc / B1, ib = loc % B1;
            (ni ? gr.vf_w2 : gr.w2)[(mb * 32 + m) * H1 + ib * 32 + nn] = v;
        } else {                                   // dW1[ib*32 + m][k = n], k == OD: db1
            const int ib = loc - B2 * B1;
            if (nn < OD) (ni ? gr.vf_w1 : gr.w1)[(ib * 32 + m) * OD + nn] = v;
            else if (nn == OD) (ni ? gr.vf_b1 : gr.b1)[ib * 32 + m] = v;
        }
    } else if (e < kOffScal) {                     // per-feature sums [net][5][64]: dW3^T[f][o] (o < 3: action head, o == 3: value head), db2[f]
        const int idx = e - kOffFeat;
        const int ni = idx / kFeatNet, o = (idx % kFeatNet) >> 6, f = idx & 63;
        if (ni >= NETS) return;
        if (NETS == 1) {
            if (o < 3) gr.pi_w3[o * H2 + f] = v;
            else if (o == 3) gr.vf_w3[f] = v;
            else gr.b2[f] = v;
        } else if (ni == 0) {
            if (o < 3) gr.pi_w3[o * H2 + f] = v;
            else if (o == 4) gr.b2[f] = v;
        } else {
            if (o == 3) gr.vf_w3[f] = v;
            else if (o == 4) gr.vf_b2[f] = v;
        }
    } else {
        const int k = e - kOffScal;
        if (k < 3) gr.pi_b3[k] = v;
        else if (k == 3) gr.vf_b3[0] = v;
        else if (k < 7) gr.log_std[k - 4] = v - ent_coef;   // entropy = sum_i (0.5 + 0.5 ln 2pi + log_std_i): d(ent_coef * -ent) = -ent_coef
        // stats[8 + j] is the running sum of stats[j] over the launches since the caller last zeroed it (one writer per
        // entry and launches are stream-ordered: no atomics) -- a training loop reads its means once per update, no add kernel
        else if (k == 7) { stats[0] = v * inv_n; stats[8] += v * inv_n; }       // pg loss
        else if (k == 8) { stats[1] = v * inv_n; stats[9] += v * inv_n; }       // value loss
        else if (k == 9) { stats[3] = v * inv_n; stats[11] += v * inv_n; }      // clip fraction
        else if (k == 10) {
            const float ent = 3.0f * 1.4189385332046727f + log_std[0] + log_std[1] + log_std[2];
            stats[4] = v * inv_n;                           // approx kl
            stats[12] += v * inv_n;
            stats[2] = ent;                                 // entropy
            stats[10] += ent;
            stats[5] = n_samples;
            stats[13] += n_samples;
        }
    }
}

template <int NETS, int H1, int H2, int OD>
int launch(const R6PolicyNet& net, const Args& a, const R6PolicyGrad& gr, float* stats, float* scratch, size_t scratch_floats,
           hipStream_t s) {
    auto kernel = ppo_grad_kernel<NETS, H1, H2, OD>;
#ifdef R6U_STAMPS
    constexpr size_t lds_bytes = sizeof(float) * (Image<NETS, H1, H2>::kTotal + kStampSlots);
#else
    constexpr size_t lds_bytes = sizeof(float) * Image<NETS, H1, H2>::kTotal;
#endif
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static std::atomic<bool> configured[16];
    const int di = (dev >= 0 && dev < 16) ? dev : 0;
    if (!configured[di].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            r6_internal_set_error("r6_ppo_minibatch_grad: cannot reserve LDS for the weight images");
            return R6_E_BADARG;
        }
        configured[di].store(dev == di, std::memory_order_relaxed);
    }
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    const int64_t n_tiles = (a.n + 31) / 32;
    int64_t blocks = (n_tiles + kWaves - 1) / kWaves;
    if (blocks > cus) blocks = cus;
    if ((size_t)blocks * kPartial > scratch_floats) {
        r6_internal_set_error("r6_ppo_minibatch_grad: scratch too small (r6_ppo_scratch_floats)");
        return R6_E_BADARG;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, net, a, scratch);
    hipLaunchKernelGGL((ppo_grad_reduce_kernel<NETS, H1, H2, OD>), dim3((kPartial + 63) / 64), dim3(1024), 0, s, scratch, (int)blocks, gr,
                       net.log_std, a.ent_coef, 1.0f / (float)a.n, (float)a.n, stats);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

// Mean and 1 / (std + 1e-8) (unbiased std, as torch.std) of adv[index[0..n)]: SB3 normalises the advantages per minibatch
// (ppo.py: `advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)`).  Two small launches: per-workgroup
// sums of x and x^2 in float64 (128 workgroups: a single one spends 158 us on the dependent index -> value loads of 65 536
// samples), then one wave folds the partial sums.
constexpr int kNormBlocks = 128;
// (blockIdx.y = minibatch b of a batched call: rows index[b * n .. (b + 1) * n), partial sums part[b][..], result out[b][2])
__global__ __launch_bounds__(256) void adv_norm_partial_kernel(const float* __restrict__ adv, const int64_t* __restrict__ index, const int64_t n,
                                                                double* __restrict__ part) {
    __shared__ double r1[4], r2[4];
    double s1 = 0.0, s2 = 0.0;
    const int64_t first = (int64_t)blockIdx.y * n;
    part += (int64_t)blockIdx.y * 2 * gridDim.x;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = (double)adv[index ? index[first + i] : first + i];
        s1 += x;
        s2 += x * x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        r1[threadIdx.x >> 6] = s1;
        r2[threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x + 0] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        part[2 * blockIdx.x + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]);
    }
}
__global__ __launch_bounds__(64) void adv_norm_final_kernel(const double* __restrict__ part, const int n_part, const int64_t n, float* __restrict__ out) {
    part += (int64_t)blockIdx.x * 2 * n_part;
    out += 2 * blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < n_part; k += 64) {
        s1 += part[2 * k];
        s2 += part[2 * k + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if (threadIdx.x == 0) {
        const double m = s1 / (double)n;
        const double var = (n > 1) ? fmax(s2 - (double)n * m * m, 0.0) / (double)(n - 1) : 0.0;
        out[0] = (float)m;
        out[1] = (float)(1.0 / (sqrt(var) + 1e-8));
    }
}

// A pseudo-random permutation of 0..n-1 without a sort (r6_permutation): out[i] = P(i), P a keyed bijection of [0, 2^bits)
// (bits = ceil(log2 n)) restricted to [0, n) by cycle walking -- values that land in [n, 2^bits) are pushed through P again,
// fewer than two applications per element on average.  P alternates steps that are each invertible modulo 2^bits: xor with a
// round key, multiplication by an odd round constant (carries information upwards), xor with the value shifted right (brings
// it back down), in the manner of Kensler's `permute` (Correlated Multi-Jittered Sampling, 2013).  torch.randperm sorts
// random keys instead: 0.22 ms for 2^20 elements on this GPU, a seventh of a PPO epoch of sixteen fused minibatch steps.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void permutation_kernel(int64_t* __restrict__ out, const int64_t n, const uint64_t seed, const uint64_t counter,
                                                           const uint64_t* __restrict__ counter_base) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = counter + (counter_base ? *counter_base : (uint64_t)0);
    const int bits = (n > 2) ? 64 - __builtin_clzll((uint64_t)(n - 1)) : 1;
    const uint64_t mask = (bits >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << bits) - 1);
    const int s1 = (bits + 1) / 2, s2 = (bits > 3) ? bits / 3 : 1;
    uint64_t key[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) key[r] = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(r + 1) + mix64(c + (uint64_t)r * 0xD1342543DE82EF95ull));
    uint64_t x = (uint64_t)i;
    do {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            x ^= key[r] & mask;
            x = (x * ((key[r] >> 17) | 1)) & mask;
            x ^= x >> ((r & 1) ? s2 : s1);
        }
    } while (x >= (uint64_t)n);
    out[i] = (int64_t)x;
}

// GAE(lambda) of a rollout, one lane per env walking its T steps backwards (r6_gae; SB3 RolloutBuffer.compute_returns_and_advantage).
// The recurrence is serial in t but the loads are not: the unrolled loop keeps eight steps of (reward, value, done) in flight.
__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                   const uint8_t* __restrict__ dones, const float* __restrict__ last_values, const int64_t T,
                                                   const int64_t n, const float gamma, const float lam, const uint8_t* __restrict__ flags,
                                                   const int flag_mask, const float* __restrict__ boot, float* __restrict__ adv,
                                                   float* __restrict__ ret) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float next_v = last_values[i], last = 0.0f;
    const bool tl = flags != nullptr && boot != nullptr;
#pragma unroll 8
    for (int64_t t = T - 1; t >= 0; --t) {
        const int64_t e = t * n + i;
        float r = rewards[e];
        const float v = values[e];
        const float nonterm = dones[e] ? 0.0f : 1.0f;
        if (tl && (flags[e] & flag_mask)) r = fmaf(gamma, boot[e], r);   // time-limit bootstrap: gamma * V(terminal observation)
        const float delta = fmaf(gamma * next_v, nonterm, r) - v;
        last = fmaf(gamma * lam * nonterm, last, delta);
        adv[e] = last;
        ret[e] = last + v;
        next_v = v;
    }
}

// torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step() (SB3 1.6's optimiser step:
// max_grad_norm 0.5, Adam eps 1e-5) for a policy whose parameters, gradients and Adam moments are flat arrays: one
// workgroup, one launch instead of ~25 (per-tensor norms, the norm of norms, the scale, three multi-tensor Adam kernels,
// the log-std clamp), which is what a PPO minibatch step spends half its GPU time on once the gradient is one launch.
__global__ __launch_bounds__(1024) void adam_clip_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                                         float* __restrict__ v, const int64_t n, const float max_norm, const float lr_value,
                                                         const float* __restrict__ lr_dev, const float beta1, const float beta2, const float eps,
                                                         int64_t* __restrict__ step_dev, const int64_t clamp_off, const int64_t clamp_n,
                                                         const float clamp_min) {
    __shared__ double red[16];
    __shared__ float coef_s;
    double ss = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) ss += (double)grad[i] * (double)grad[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        const float total_norm = (float)sqrt(t);
        coef_s = (max_norm > 0.0f) ? fminf(max_norm / (total_norm + 1e-6f), 1.0f) : 1.0f;   // clip_grad_norm_: clamped to 1
    }
    __syncthreads();
    const float coef = coef_s;
    const int64_t t = *step_dev + 1;                       // torch.optim.Adam: step count starts at 1
    const float lr = lr_dev ? *lr_dev : lr_value;
    const float bc1 = 1.0f - powf(beta1, (float)t), bc2 = 1.0f - powf(beta2, (float)t);
    const float step_size = lr / bc1, isq_bc2 = 1.0f / sqrtf(bc2);
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float g = grad[i] * coef;
        const float mi = fmaf(beta1, m[i], (1.0f - beta1) * g);        // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = fmaf(beta2, v[i], (1.0f - beta2) * g * g);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        float p = param[i] - step_size * mi / fmaf(sqrtf(vi), isq_bc2, eps);
        if (i >= clamp_off && i < clamp_off + clamp_n) p = fmaxf(p, clamp_min);
        param[i] = p;
    }
    __syncthreads();
    if (threadIdx.x == 0) *step_dev = t;
}

// The same step for n <= K * 1024 (the policies here have ~10.4 k parameters): every thread fetches its K elements of grad, m, v
// and param up front -- 4 K loads in flight, ONE memory round trip -- and keeps them in registers across the norm reduction.
// The strided loops above pay a dependent round trip per iteration: 10.4 us for 10 311 parameters, a ninth of a fused PPO
// minibatch step.
template <int K>
__global__ __launch_bounds__(1024) void adam_clip_small_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                                               float* __restrict__ v, const int n, const float max_norm, const float lr_value,
                                                               const float* __restrict__ lr_dev, const float beta1, const float beta2,
                                                               const float eps, int64_t* __restrict__ step_dev, const int clamp_off,
                                                               const int clamp_n, const float clamp_min) {
    __shared__ float red[16];
    float g[K], mm[K], vv[K], pp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = k * 1024 + threadIdx.x;
        const bool in = i < n;
        g[k] = in ? grad[i] : 0.0f;
        mm[k] = in ? m[i] : 0.0f;
        vv[k] = in ? v[i] : 0.0f;
        pp[k] = in ? param[i] : 0.0f;
    }
    const int64_t t = *step_dev + 1;                       // torch.optim.Adam: step count starts at 1
    const float lr = lr_dev ? *lr_dev : lr_value;
    // the norm as torch forms it (fp32 squares): a lane's K squares and the wave's sum in fp32, the 16 wave sums in double by
    // EVERY thread after one barrier -- no serial section of one thread, no second barrier (round 2: double shuffles, thread 0
    // adding and taking a double square root between two barriers: 8.4 us per step for 10 311 parameters)
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) ss = fmaf(g[k], g[k], ss);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    // the bias corrections meanwhile: beta^t = 2^(t log2 beta) on the transcendental unit
    const float tf = (float)t;
    const float bc1 = 1.0f - __builtin_amdgcn_exp2f(tf * __log2f(beta1)), bc2 = 1.0f - __builtin_amdgcn_exp2f(tf * __log2f(beta2));
    const float step_size = lr / bc1, isq_bc2 = __builtin_amdgcn_rsqf(bc2);
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += (double)red[k];
    const float total_norm = sqrtf((float)tot);
    const float coef = (max_norm > 0.0f) ? fminf(max_norm / (total_norm + 1e-6f), 1.0f) : 1.0f;   // clip_grad_norm_: clamped to 1
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = k * 1024 + threadIdx.x;
        if (i < n) {
            const float gi = g[k] * coef;
            const float mi = fmaf(beta1, mm[k], (1.0f - beta1) * gi);
            const float vi = fmaf(beta2, vv[k], (1.0f - beta2) * gi * gi);
            m[i] = mi;
            v[i] = vi;
            float p = pp[k] - step_size * mi / fmaf(sqrtf(vi), isq_bc2, eps);
            if (i >= clamp_off && i < clamp_off + clamp_n) p = fmaxf(p, clamp_min);
            param[i] = p;
        }
    }
    if (threadIdx.x == 0) *step_dev = t;
}

// The same step spread over ceil(n / 1024) workgroups: every workgroup forms the WHOLE norm itself (K gradient loads per thread,
// the same fixed-order sums as above, so all workgroups get the same bits) and then updates its own 1024 elements -- one element
// of m, v, param per thread instead of K: the loads and stores of the optimiser state, the longer part of the one-workgroup
// form, run on ceil(n / 1024) CUs at once.  Nobody reads what another workgroup writes (gradients are read-only here).  The step
// count is read and advanced without a second buffer: every workgroup adds 2^48 to the word (an arrival count in bits the step
// never reaches) and takes the step from the value the add returned; the workgroup that arrives last knows every other has
// read it and stores the new count, arrival bits cleared.
template <int K>
__global__ __launch_bounds__(1024) void adam_clip_wide_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                                              float* __restrict__ v, const int n, const float max_norm, const float lr_value,
                                                              const float* __restrict__ lr_dev, const float beta1, const float beta2,
                                                              const float eps, int64_t* __restrict__ step_dev, const int clamp_off,
                                                              const int clamp_n, const float clamp_min) {
    constexpr unsigned long long kArrive = 1ull << 48;
    __shared__ float red[16];
    __shared__ unsigned long long old_s;
    if (threadIdx.x == 0) old_s = atomicAdd(reinterpret_cast<unsigned long long*>(step_dev), kArrive);
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool in = i < n;
    const int ii = in ? i : 0;
    const float mi0 = m[ii], vi0 = v[ii], pi0 = param[ii];
    float g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = k * 1024 + threadIdx.x;
        const float gj = grad[j < n ? j : 0];
        g[k] = j < n ? gj : 0.0f;
    }
    const float lr = lr_dev ? *lr_dev : lr_value;
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) ss = fmaf(g[k], g[k], ss);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    float gi = 0.0f;   // this thread's own gradient: element k * 1024 + threadIdx.x with k = blockIdx.x
#pragma unroll
    for (int k = 0; k < K; ++k) gi = (k == (int)blockIdx.x) ? g[k] : gi;
    __syncthreads();
    const unsigned long long old = old_s;
    const int64_t t = (int64_t)(old & (kArrive - 1)) + 1;   // torch.optim.Adam: step count starts at 1
    const float tf = (float)t;
    const float bc1 = 1.0f - __builtin_amdgcn_exp2f(tf * __log2f(beta1)), bc2 = 1.0f - __builtin_amdgcn_exp2f(tf * __log2f(beta2));
    const float step_size = lr / bc1, isq_bc2 = __builtin_amdgcn_rsqf(bc2);
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += (double)red[k];
    const float total_norm = sqrtf((float)tot);
    const float coef = (max_norm > 0.0f) ? fminf(max_norm / (total_norm + 1e-6f), 1.0f) : 1.0f;   // clip_grad_norm_: clamped to 1
    if (in) {
        gi *= coef;
        const float mi = fmaf(beta1, mi0, (1.0f - beta1) * gi);
        const float vi = fmaf(beta2, vi0, (1.0f - beta2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        float p = pi0 - step_size * mi / fmaf(sqrtf(vi), isq_bc2, eps);
        if (i >= clamp_off && i < clamp_off + clamp_n) p = fmaxf(p, clamp_min);
        param[i] = p;
    }
    if (threadIdx.x == 0 && (old >> 48) + 1 == gridDim.x) *step_dev = t;
}

}  // namespace r6u

extern "C" int r6_adam_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float max_grad_norm,
                                 float lr, const float* lr_dev, float beta1, float beta2, float eps, int64_t* step_dev,
                                 int64_t clamp_off, int64_t clamp_n, float clamp_min, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n <= 0) {
        r6_internal_set_error("r6_adam_clip_step: param, grad, exp_avg, exp_avg_sq, step_dev and n > 0 are required");
        return R6_E_BADARG;
    }
    constexpr int kSmall = 12;
    static const int wide = [] { const char* e = std::getenv("R6_ADAM_WIDE"); return e ? std::atoi(e) : 1; }();   // A/B: 0 = the one-workgroup form
    if (n <= kSmall * 1024 && wide)
        hipLaunchKernelGGL(r6u::adam_clip_wide_kernel<kSmall>, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                           param, grad, exp_avg, exp_avg_sq, (int)n, max_grad_norm, lr, lr_dev, beta1, beta2, eps, step_dev, (int)clamp_off,
                           (int)clamp_n, clamp_min);
    else if (n <= kSmall * 1024)
        hipLaunchKernelGGL(r6u::adam_clip_small_kernel<kSmall>, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), param, grad, exp_avg,
                           exp_avg_sq, (int)n, max_grad_norm, lr, lr_dev, beta1, beta2, eps, step_dev, (int)clamp_off, (int)clamp_n, clamp_min);
    else
        hipLaunchKernelGGL(r6u::adam_clip_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, n,
                           max_grad_norm, lr, lr_dev, beta1, beta2, eps, step_dev, clamp_off, clamp_n, clamp_min);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

extern "C" int r6_minibatch_adv_norm(const float* adv, const int64_t* index, int64_t n, int64_t n_batches, float* out2, double* scratch,
                                     void* stream) {
    if (!adv || !out2 || !scratch || n <= 0 || n_batches <= 0 || n_batches > 65535) {
        r6_internal_set_error("r6_minibatch_adv_norm: adv, out, scratch (256 doubles per minibatch), n > 0 and 1 <= n_batches <= 65535 are required");
        return R6_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = (int)((n + 255) / 256);
    if (blocks > r6u::kNormBlocks) blocks = r6u::kNormBlocks;
    hipLaunchKernelGGL(r6u::adv_norm_partial_kernel, dim3(blocks, (unsigned)n_batches), dim3(256), 0, s, adv, index, n, scratch);
    hipLaunchKernelGGL(r6u::adv_norm_final_kernel, dim3((unsigned)n_batches), dim3(64), 0, s, scratch, blocks, n, out2);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

extern "C" int r6_permutation(int64_t* out, int64_t n, uint64_t seed, uint64_t counter, const uint64_t* counter_base, void* stream) {
    if (!out || n < 0) {
        r6_internal_set_error("r6_permutation: out and n >= 0 are required");
        return R6_E_BADARG;
    }
    if (n == 0) return 0;
    hipLaunchKernelGGL(r6u::permutation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, seed,
                       counter, counter_base);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

extern "C" int r6_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, int64_t T, int64_t n,
                      float gamma, float lam, const uint8_t* flags, int flag_mask, const float* boot_values, float* adv, float* ret,
                      void* stream) {
    if (!rewards || !values || !dones || !last_values || !adv || !ret || T < 0 || n < 0) {
        r6_internal_set_error("r6_gae: rewards, values, dones, last_values, adv and ret are required");
        return R6_E_BADARG;
    }
    if (T == 0 || n == 0) return 0;
    hipLaunchKernelGGL(r6u::gae_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rewards, values, dones,
                       last_values, T, n, gamma, lam, flags, flag_mask, boot_values, adv, ret);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

extern "C" size_t r6_ppo_scratch_floats(void) { return (size_t)256 * r6u::kPartial; }
extern "C" size_t r6_policy_grad_size(void) { return sizeof(R6PolicyGrad); }

extern "C" int r6_ppo_minibatch_grad(const R6PolicyNet* net, const float* obs, const float* action, const float* old_logp,
                                     const float* adv, const float* ret, const int64_t* index, int64_t n, int64_t n_rows,
                                     const float* adv_norm, float clip_range, float vf_coef, float ent_coef, const R6PolicyGrad* grad,
                                     float* stats, float* scratch, size_t scratch_floats, void* stream) {
    if (!net || !obs || !action || !old_logp || !adv || !ret || !grad || !stats || !scratch) {
        r6_internal_set_error("r6_ppo_minibatch_grad: all pointers except index and adv_norm are required");
        return R6_E_BADARG;
    }
    if (n <= 0 || n >= ((int64_t)1 << 31)) {
        r6_internal_set_error("r6_ppo_minibatch_grad: need 0 < n < 2^31");
        return R6_E_BADARG;
    }
    // the kernel keeps the gathered row numbers as 32-bit values (one tile ahead of their use): a rollout buffer with 2^31 rows
    // or more (e.g. 2^20 envs x 2048 steps) would silently gather the wrong rows
    if (n_rows < n || (index && n_rows >= ((int64_t)1 << 31))) {
        r6_internal_set_error("r6_ppo_minibatch_grad: need n <= n_rows, and n_rows < 2^31 when an index is given");
        return R6_E_BADARG;
    }
    const R6PolicyNet& p = *net;
    const R6PolicyGrad& g = *grad;
    const bool trunk = p.shared && p.hidden1 == 128 && p.hidden2 == 64;      // the reference's net_arch=[128, 64]
    const bool two = !p.shared && p.hidden1 == 64 && p.hidden2 == 64;        // SB3's default: separate (64, 64) MLPs
    if (!(trunk || two) || (p.obs_dim != 13 && p.obs_dim != 14)) {
        r6_internal_set_error("r6_ppo_minibatch_grad: supported nets: one shared trunk (128, 64) or two separate MLPs (64, 64); obs_dim 13|14");
        return R6_E_BADARG;
    }
    if (!p.pi_w1 || !p.pi_b1 || !p.pi_w2 || !p.pi_b2 || !p.pi_w3 || !p.pi_b3 || !p.vf_w3 || !p.vf_b3 || !p.log_std ||
        !g.w1 || !g.b1 || !g.w2 || !g.b2 || !g.pi_w3 || !g.pi_b3 || !g.vf_w3 || !g.vf_b3 || !g.log_std) {
        r6_internal_set_error("r6_ppo_minibatch_grad: trunk / head pointers of R6PolicyNet and of R6PolicyGrad are required");
        return R6_E_BADARG;
    }
    if (two && (!p.vf_w1 || !p.vf_b1 || !p.vf_w2 || !p.vf_b2 || !g.vf_w1 || !g.vf_b1 || !g.vf_w2 || !g.vf_b2)) {
        r6_internal_set_error("r6_ppo_minibatch_grad: separate MLPs need vf_w1, vf_b1, vf_w2, vf_b2 in R6PolicyNet and R6PolicyGrad");
        return R6_E_BADARG;
    }
    r6u::Args a{obs, action, old_logp, adv, ret, index, adv_norm, n, clip_range, vf_coef, ent_coef};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (trunk) {
        if (p.obs_dim == 13) return r6u::launch<1, 128, 64, 13>(p, a, g, stats, scratch, scratch_floats, s);
        return r6u::launch<1, 128, 64, 14>(p, a, g, stats, scratch, scratch_floats, s);
    }
    if (p.obs_dim == 13) return r6u::launch<2, 64, 64, 13>(p, a, g, stats, scratch, scratch_floats, s);
    return r6u::launch<2, 64, 64, 14>(p, a, g, stats, scratch, scratch_floats, s);
}
