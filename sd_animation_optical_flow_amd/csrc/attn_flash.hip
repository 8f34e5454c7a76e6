// Fused attention for the UNet's head sizes: out = softmax(q k^T * scale + bias) v without the
// score matrix ever leaving the CU -- the form `xformers.ops.memory_efficient_attention` has at
// ldm/modules/attention.py:314,426 (16384 tokens x 8 heads at 1024x1024: 8.6 GB of scores per image in the unfused form).
//
// Written for the 32x32x2 fp32 matrix-core instruction, transposed so that nothing has to move between lanes:
//
//   S^T = K Q^T     A operand = K tile rows from LDS (b128 reads: a lane takes 4 consecutive d of its key),
//                   B operand = the wave's 32 queries, pre-scaled by scale * log2(e), held in registers for the whole key loop.
//                   The accumulator of lane (c = lane % 32, h = lane / 32) holds S^T[key = 8i + 4h + j][query = c]:
//                   one query per lane column, so the row maximum / row sum of the softmax are IN-LANE reductions over 16
//                   registers plus one exchange with lane ^ 32.
//   O^T += V^T P^T  the k index of this product is the key; the instruction wants, from half h of the wave, the key pair
//                   member h -- and register 4i + j of half h already holds key 8i + 4h + j.  So MFMA number 4i + j takes P^T
//                   straight from the accumulator register it was computed in, with the V operand read as
//                   V[key = 8i + 4h + j][d = c]: no shuffle, no LDS round trip for P.
//
// Online softmax in base 2 (v_exp_f32): m = running maximum, l = running sum per lane half (the halves share m, their partial
// sums are added once at the end), O rescaled only when some query's maximum moved (wave-uniform branch; exact, the factor
// is 1.0 otherwise).  K / V tiles of 64 (32 for d > 40) keys are double-buffered in LDS behind register-staged global loads.
// Everything is fp32; summation order differs from the unfused path, results agree to rounding.
#include "ofx_internal.h"

#include <cmath>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

struct FlashArgs {
    const float* q;
    const float* k;
    const float* v;
    const float* bias;   // [Nq, Nk] per batch-head (bias_bs = Nq * Nk) or shared (bias_bs = 0); may be null
    float* out;
    long bias_bs;
    int BH, Nq, Nk;
    float scale_log2e;
    int qtiles;
    int xcd_grouped;
};

constexpr float kLog2e = 1.4426950408889634f;

template <int D, int BK>
__global__ __launch_bounds__(256) void flash_attn_kernel(const FlashArgs a) {
    static_assert(D % 8 == 0 && D <= 160, "head size");
    constexpr int DQ = D / 8;                       // groups of 8 along d: each wave half takes 4 of them
    constexpr int DT = (D + 31) / 32;               // 32-row tiles of O^T
    constexpr int LDK = D + 4;                      // K row stride (floats): conflict-free b128 fragment reads
    constexpr int LDV = ((D + 7) / 16) * 16 + 8;    // V row stride: = 8 mod 16, so keys 4 apart sit 32 banks apart
    constexpr int KT = BK * LDK;
    constexpr int VT = BK * LDV + 32;               // slack: the last d tile reads past D (into rows that are never stored)
    constexpr int NF4 = BK * D / 4;                 // float4 per tensor per tile
    constexpr int PER = (NF4 + 255) / 256;
    extern __shared__ float smem[];                 // [2][KT + VT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;

    int bh, qt;
    if (a.xcd_grouped) {
        // consecutive workgroup ids go round the 8 XCDs: give every XCD whole batch-heads so that their K / V stay in its L2
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        bh = (slot / a.qtiles) * 8 + xcd;
        qt = slot % a.qtiles;
    } else {
        bh = blockIdx.x / a.qtiles;
        qt = blockIdx.x % a.qtiles;
    }
    const int qrow = qt * 128 + wave * 32 + c;
    const bool qok = qrow < a.Nq;

    // the wave's queries, scaled: lane (c, h) keeps q[qrow][8m + 4h + 0..3]
    float4 qreg[DQ];
    {
        const float* qp = a.q + ((long)bh * a.Nq + (qok ? qrow : 0)) * D + 4 * h;
#pragma unroll
        for (int m = 0; m < DQ; ++m) {
            float4 t = qok ? *reinterpret_cast<const float4*>(qp + 8 * m) : make_float4(0.f, 0.f, 0.f, 0.f);
            qreg[m] = make_float4(t.x * a.scale_log2e, t.y * a.scale_log2e, t.z * a.scale_log2e, t.w * a.scale_log2e);
        }
    }

    const float* kg = a.k + (long)bh * a.Nk * D;
    const float* vg = a.v + (long)bh * a.Nk * D;
    const float* bg = a.bias ? a.bias + (long)bh * a.bias_bs + (long)(qok ? qrow : 0) * a.Nk : nullptr;

    float4 sk[PER], sv[PER];
    auto load_tile = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + 256 * p;
            const int key = f / (D / 4);
            const bool ok = (PER * 256 == NF4 || f < NF4) && k0 + key < a.Nk;
            // a tile is one contiguous run of BK * D floats
            sk[p] = ok ? *reinterpret_cast<const float4*>(kg + (long)k0 * D + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[p] = ok ? *reinterpret_cast<const float4*>(vg + (long)k0 * D + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = tid + 256 * p;
            if (PER * 256 == NF4 || f < NF4) {
                const int key = f / (D / 4), c4 = f % (D / 4);
                *reinterpret_cast<float4*>(buf + key * LDK + 4 * c4) = sk[p];
                *reinterpret_cast<float4*>(buf + KT + key * LDV + 4 * c4) = sv[p];
            }
        }
    };

    v16f o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (a.Nk + BK - 1) / BK;
    load_tile(0);
    store_tile(smem);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const float* buf = smem + (t & 1) * (KT + VT);
        if (t + 1 < nt) load_tile((t + 1) * BK);
        const int k0 = t * BK;
#pragma unroll
        for (int kb = 0; kb < BK / 32; ++kb) {
            const int kbase = k0 + kb * 32;
            if (kbase >= a.Nk) break;                       // wave-uniform
            v16f s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const float* kr = buf + (kb * 32 + c) * LDK + 4 * h;
#pragma unroll
            for (int m = 0; m < DQ; ++m) {
                const float4 kf = *reinterpret_cast<const float4*>(kr + 8 * m);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[m].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[m].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[m].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[m].w, s, 0, 0, 0);
            }
            if (bg) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = kbase + 8 * i + 4 * h + j;
                        if (key < a.Nk) s[4 * i + j] += bg[key] * kLog2e;
                    }
            }
            if (kbase + 32 > a.Nk) {                        // ragged last block: keys past the end take no weight
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kbase + 8 * i + 4 * h + j >= a.Nk) s[4 * i + j] = -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;       // a row whose every key so far is masked out
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // exp2(-inf) = 0 on the first block
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = __builtin_amdgcn_exp2f(s[e] - m_use);
                ps += s[e];
            }
            l_run = l_run * alpha + ps;
            if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
                for (int tt = 0; tt < DT; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[tt][e] *= alpha;
            }
            m_run = m_new;
            const float* vr = buf + KT + (kb * 32 + 4 * h) * LDV + c;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt)
                        o[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[(8 * i + j) * LDV + 32 * tt], s[4 * i + j], o[tt], 0, 0, 0);
                }
        }
        if (t + 1 < nt) store_tile(smem + ((t + 1) & 1) * (KT + VT));
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qok) {
        float* op = a.out + ((long)bh * a.Nq + qrow) * D;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d0 = 32 * tt + 8 * i + 4 * h;
                if (d0 < D)
                    *reinterpret_cast<float4*>(op + d0) =
                        make_float4(o[tt][4 * i] * inv, o[tt][4 * i + 1] * inv, o[tt][4 * i + 2] * inv, o[tt][4 * i + 3] * inv);
            }
    }
}

template <int D, int BK>
int launch_flash(const FlashArgs& a, hipStream_t s) {
    constexpr int LDK = D + 4, LDV = ((D + 7) / 16) * 16 + 8;
    constexpr size_t lds = 2 * (size_t)(BK * LDK + BK * LDV + 32) * sizeof(float);
    if (lds > 65536) {                               // > 64 KB of dynamic LDS needs the opt-in (per device: set on every launch, it is cheap)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_kernel<D, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((flash_attn_kernel<D, BK>), dim3((unsigned)(a.BH * a.qtiles)), dim3(256), lds, s, a);
    return ofx_launch_status();
}

}  // namespace

bool ofx_attention_flash_ok(int D) { return D == 40 || D == 64 || D == 80 || D == 128 || D == 160; }

int ofx_attention_flash_launch(const float* q, const float* k, const float* v, const float* bias, long bias_bstride, float* out, int BH, int Nq,
                               int Nk, int D, float scale, hipStream_t s) {
    FlashArgs a{};
    a.q = q; a.k = k; a.v = v; a.bias = bias; a.out = out;
    a.bias_bs = bias_bstride;
    a.BH = BH; a.Nq = Nq; a.Nk = Nk;
    a.scale_log2e = scale * kLog2e;
    a.qtiles = ofx_cdiv(Nq, 128);
    a.xcd_grouped = (BH % 8 == 0) ? 1 : 0;
    if ((long)BH * a.qtiles > 0x7fffffffL) return OFX_EINVAL;
    OfxProfScope prof("attn_flash", s);
    prof.flops(4.0 * BH * (double)Nq * Nk * D);
    switch (D) {
        case 40: return launch_flash<40, 64>(a, s);
        case 64: return launch_flash<64, 32>(a, s);
        case 80: return launch_flash<80, 32>(a, s);
        case 128: return launch_flash<128, 32>(a, s);
        case 160: return launch_flash<160, 32>(a, s);
    }
    return OFX_EINVAL;
}
