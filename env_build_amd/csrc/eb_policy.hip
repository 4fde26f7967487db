// eb_policy.hip — the policy network in the loop, gfx950 only.
//
// MLPNet of the reference (utils/model.py:18-43): Dense(obs_dim -> U, act), (L-1) x Dense(U -> U, act),
// Dense(U -> out_dim, out_act), fp32, evaluated for a batch of observations in ONE launch.  This is the only
// dense contraction on the path, so it is the only kernel on the matrix cores:
//
//   * v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate.  It is exact fp32 and accumulates as a chain of fused
//     multiply-adds in k order, so the CPU oracle reproduces every bit with fmaf (include/envbuild.h states
//     the contract).  Peak 157 TFLOP/s (MI355X_MICROARCH.md): 64 cycles per instruction per SIMD.
//   * one block = 64 rows (observations) x 4 waves.  The activations of the current layer live in LDS
//     (64 x K floats, one buffer: a layer's outputs stay in the accumulators until every wave has finished
//     reading its inputs); the weights stream from L2 (all layers together are < 1 MB, shared by every block).
//   * operand layouts are chosen so that every fragment load is one 16-byte access per lane feeding FOUR
//     MFMAs: lane l of a 32x32x2 MFMA holds A[i = l & 31][k = 2m + (l >> 5)], so LDS keeps row i as
//     [k even | k odd] halves (m consecutive -> ds_read_b128 = 4 k-pairs); the host packs W the same way per
//     32-column tile (eb_mlp_set_layer -> pack_weights), so a wave's global_load_dwordx4 is 1 KB contiguous.
//   * a wave owns RT x CT tiles of 32 x 32 outputs (U = 256: 2 x 2; 512: 2 x 4; 128: 2 x 1; 64: 1 x 1), i.e.
//     RT + CT fragment loads per 4 * RT * CT MFMAs; the output layer (<= 32 columns) runs on 16 x 16 x 4 tiles, one row
//     tile of 16 per wave (all four waves).
//   * widths are padded with zero weights / zero bias to the next supported U; a padded unit outputs
//     act(0) = 0 for all four activations and meets zero weights in the next layer — exact no-ops in the chain.
#include "eb_kernels.h"

namespace eb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- deterministic exp / tanh (Cephes scheme, explicit fma; oracle: eb_expf / eb_tanhf) ----
// Written without branches (selects): the epilogue applies them to 64 accumulator values per lane and layer.
EB_DEV float exp_det(float x0) {
    const float x = x0 > 88.0f ? 88.0f : (x0 < -87.0f ? -87.0f : x0);   // NaN falls through both compares
    const float fx = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(-fx, 0.693359375f, x);
    r = __builtin_fmaf(-fx, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, z, r) + 1.0f;
    const int n = (x0 == x0) ? (int)fx : 0;                              // -126 .. 127
    const float v = y * __builtin_bit_cast(float, (unsigned)(n + 127) << 23);
    return (x0 == x0) ? v : x0;
}

EB_DEV float tanh_det(float x) {
    const float ax = __builtin_fabsf(x);
    const float s = exp_det(ax + ax);
    const float t = 1.0f - 2.0f / (s + 1.0f);
    const float big = x < 0.0f ? -t : t;
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, x, x);
    const float sat = x > 0.0f ? 1.0f : -1.0f;
    return ax > 44.0f ? sat : (ax >= 0.625f ? big : small);              // NaN: both compares false -> small = NaN
}

// ELU = x > 0 ? x : exp_det(x) - 1, written for the epilogue (64 values per lane and layer): only x <= 0 reaches the
// exponential's result, so the upper clamp goes, the scaling by 2^n is one v_ldexp_f32 (exactly the multiplication by the
// bit-built power of two, denormal results included) and a NaN needs no select — it travels through the fma chain by
// itself (v_cvt_i32_f32 of NaN is 0).  Same bits as exp_det(x) - 1 for every x <= 0, so the oracle's eb_expf stays as it is.
EB_DEV float elu_det(float x0) {
    const float x = x0 < -87.0f ? -87.0f : x0;
    const float fx = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(-fx, 0.693359375f, x);
    r = __builtin_fmaf(-fx, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, z, r) + 1.0f;
    const float v = __builtin_amdgcn_ldexpf(y, (int)fx);
    return x0 > 0.0f ? x0 : v - 1.0f;
}

template <int ACT>
EB_DEV float activate(float x) {
    if (ACT == MLP_ACT_RELU) return x > 0.0f ? x : 0.0f;
    if (ACT == MLP_ACT_ELU) return elu_det(x);
    if (ACT == MLP_ACT_TANH) return tanh_det(x);
    return x;
}
EB_DEV float activate_rt(int act, float x) {
    switch (act) {
        case MLP_ACT_RELU: return activate<MLP_ACT_RELU>(x);
        case MLP_ACT_ELU: return activate<MLP_ACT_ELU>(x);
        case MLP_ACT_TANH: return activate<MLP_ACT_TANH>(x);
        default: return x;
    }
}

// One layer's k-loop for the RT x CT tiles of a wave.  a_row: LDS address of A[row tile rt0][i][h][0];
// wp: this layer's packed weights; steps = k_pad / 8.  acc enters holding the bias.  Fragments are fetched two
// steps ahead into one of three register sets; the loop is unrolled by three so that the sets rotate by name
// (a register copy would make every step wait for the load it has just issued).
template <int RT, int CT>
EB_DEV void layer_chain(const float* a_row, int row_tile_stride, const f32x4* __restrict__ wp, int steps, int ct0,
                        int lane, f32x16 (&acc)[RT][CT]) {
    const f32x4* bsrc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bsrc[c] = wp + (size_t)(ct0 + c) * steps * 64 + lane;
    f32x4 bq[3][CT], aq[3][RT];
    auto fetch = [&](int set, int s) {
        const int sc = s < steps ? s : steps - 1;
#pragma unroll
        for (int c = 0; c < CT; ++c) bq[set][c] = bsrc[c][(size_t)sc * 64];
#pragma unroll
        for (int r = 0; r < RT; ++r) aq[set][r] = *reinterpret_cast<const f32x4*>(a_row + r * row_tile_stride + sc * 4);
    };
    auto run = [&](int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q)                                                          // k pairs in order
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[set][r][q], bq[set][c][q], acc[r][c], 0, 0, 0);
    };
    // the scheduling fences keep every fetch where it is written: two steps (32 MFMAs) ahead of its use
#define EB_STEP(FSET, FS, RSET)                   \
    fetch(FSET, FS);                              \
    __builtin_amdgcn_sched_barrier(0);            \
    run(RSET);                                    \
    __builtin_amdgcn_sched_barrier(0)
    fetch(0, 0);
    fetch(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    int s = 0;
    for (; s + 3 <= steps; s += 3) {
        EB_STEP(2, s + 2, 0);
        EB_STEP(0, s + 3, 1);
        EB_STEP(1, s + 4, 2);
    }
    if (s < steps) { EB_STEP(2, s + 2, 0); }
    if (s + 1 < steps) run(1);
#undef EB_STEP
}

// A layer's outputs back into the LDS activation buffer (the layout below), through the activation.
template <int RT, int CT, int ACT>
EB_DEV void store_hidden(float* lds, int RS, int HS, int rt0, int ct0, int i, int h, const f32x16 (&acc)[RT][CT]) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = (ct0 + c) * 32 + i;
            float* dst = lds + (col & 1) * HS + (col >> 1);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = (rt0 + r) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                dst[row * RS] = activate<ACT>(acc[r][c][v]);
            }
        }
}

// LDS layout of the activations: element (row i, input k) at i * RS + (k & 1) * HS + (k >> 1); RS = Kmax + 4
// floats (the pad spreads the 32 rows of a fragment read over all banks), HS = Kmax / 2.
template <int RT, int CT>
__global__ __launch_bounds__(MLP_THREADS, CT == 4 ? 1 : 2) void mlp_kernel(const MlpArgs A) {   // waves per SIMD
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int RS = A.row_stride, HS = (RS - 4) >> 1;
    const int row0 = blockIdx.x * MLP_ROWS;
    const int rows_here = A.n - row0 < MLP_ROWS ? A.n - row0 : MLP_ROWS;
    const int D = A.obs_dim, K0 = A.hid[0].k_pad;

    // ---- stage the (preprocessed) observations: a wave takes 16 rows, lanes stride over a row (coalesced);
    //      the loads of three column chunks x 16 rows are in flight together; zero beyond n and obs_dim ----
    {
        constexpr int RPW = MLP_ROWS / 4, KC = 3;
        const int rbase = wave * RPW;
        for (int k0 = lane; k0 < K0; k0 += 64 * KC) {
            float v[KC][RPW], sc[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int k = k0 + 64 * c, kc = k < D ? k : D - 1;
                sc[c] = A.scale ? A.scale[kc] : 1.0f;                     // x * 1.0f is x, bit for bit
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int r = rbase + rr;
                    const int rc = r < rows_here ? r : rows_here - 1;
                    v[c][rr] = A.obs[(size_t)(row0 + rc) * D + kc];
                }
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int k = k0 + 64 * c;
                if (k < K0) {
                    float* dst = lds + rbase * RS + (k & 1) * HS + (k >> 1);
#pragma unroll
                    for (int rr = 0; rr < RPW; ++rr)
                        dst[rr * RS] = (rbase + rr < rows_here && k < D) ? v[c][rr] * sc[c] : 0.0f;   // preprocessor.py:121
                }
            }
        }
    }
    __syncthreads();

    // ---- hidden layers ----
    const int rt0 = RT == 2 ? 0 : (wave & 1);
    const int ct0 = RT == 2 ? wave * CT : (wave >> 1);
    for (int L = 0; L < A.n_hidden; ++L) {
        const MlpLayer& ly = A.hid[L];
        f32x16 acc[RT][CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float b = ly.b[(ct0 + c) * 32 + i];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[r][c][v] = b;
        }
        layer_chain<RT, CT>(lds + (rt0 * 32 + i) * RS + h * HS, 32 * RS, reinterpret_cast<const f32x4*>(ly.w),
                            ly.k_pad >> 3, ct0, lane, acc);
        __syncthreads();                                              // every wave has read this layer's inputs
        switch (A.hidden_act) {
            case MLP_ACT_RELU: store_hidden<RT, CT, MLP_ACT_RELU>(lds, RS, HS, rt0, ct0, i, h, acc); break;
            case MLP_ACT_ELU: store_hidden<RT, CT, MLP_ACT_ELU>(lds, RS, HS, rt0, ct0, i, h, acc); break;
            case MLP_ACT_TANH: store_hidden<RT, CT, MLP_ACT_TANH>(lds, RS, HS, rt0, ct0, i, h, acc); break;
            default: store_hidden<RT, CT, MLP_ACT_LINEAR>(lds, RS, HS, rt0, ct0, i, h, acc); break;
        }
        __syncthreads();
    }

    // ---- output layer: 16 x 16 x 4 tiles (v_mfma_f32_16x16x4_f32: same exact fp32, same k-ordered fma chain, half the
    // issue time of a 32 x 32 x 2 per k), row tile = wave — all four waves, where the 32-column tile of rounds 1-2 kept two of
    // them idle for 128 MFMAs of 64 cycles — and ceil(out_dim / 16) column tiles.  Lane l supplies A[i = l & 15][k = 4s + (l >> 4)]
    // and B[k = 4s + (l >> 4)][j = l & 15]; it receives D[4 (l >> 4) + v][l & 15], v = 0..3.
    {
        const int i16 = lane & 15, kq = lane >> 4, hsel = kq >> 1;
        const int steps4 = A.outl.k_pad >> 4;                               // groups of four MFMAs (16 inputs)
        const float* a_ptr = lds + (wave * 16 + i16) * RS + (kq & 1) * HS;  // element (k >> 1) = m of this lane's parity at a_ptr[m]
        const f32x4* w16 = reinterpret_cast<const f32x4*>(A.outl.w);
        const int ct16 = (A.out_dim + 15) >> 4;
        for (int ct = 0; ct < ct16; ++ct) {
            const float b = A.outl.b[ct * 16 + i16];
            f32x4 acc = {b, b, b, b};
            const f32x4* wsrc = w16 + (size_t)ct * steps4 * 64 + lane;
            f32x4 bq = wsrc[0];
            for (int s4 = 0; s4 < steps4; ++s4) {
                const f32x4 bn = wsrc[(size_t)(s4 + 1 < steps4 ? s4 + 1 : s4) * 64];      // the next group's weights under these MFMAs
                const f32x4 a01 = *reinterpret_cast<const f32x4*>(a_ptr + 8 * s4);
                const f32x4 a23 = *reinterpret_cast<const f32x4*>(a_ptr + 8 * s4 + 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hsel ? a01[1] : a01[0], bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hsel ? a01[3] : a01[2], bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hsel ? a23[1] : a23[0], bq[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hsel ? a23[3] : a23[2], bq[3], acc, 0, 0, 0);
                bq = bn;
            }
            const int col = ct * 16 + i16;
            if (A.head == MLP_HEAD_LOGITS) {
                if (col < A.out_dim) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = wave * 16 + 4 * kq + v;
                        if (row < rows_here) A.out[(size_t)(row0 + row) * A.out_dim + col] = activate_rt(A.out_act, acc[v]);
                    }
                }
            } else {   // deterministic action: action_range * tanh(mean), utils/policy.py:89-92
                const int act_dim = A.out_dim >> 1;
                if (col < act_dim) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = wave * 16 + 4 * kq + v;
                        const float mean = activate_rt(A.out_act, acc[v]);
                        if (row < rows_here)
                            A.out[(size_t)(row0 + row) * act_dim + col] = A.action_range > 0.0f ? A.action_range * tanh_det(mean) : mean;
                    }
                }
            }
        }
    }
}

int mlp_padded_units(int n_units) {
    return n_units <= 64 ? 64 : (n_units <= 128 ? 128 : (n_units <= 256 ? 256 : 512));
}

size_t mlp_lds_bytes(const MlpArgs& A) { return (size_t)MLP_ROWS * A.row_stride * sizeof(float); }

hipError_t launch_mlp(const MlpArgs& A, hipStream_t s) {
    if (A.n <= 0) return hipSuccess;
    const dim3 g((A.n + MLP_ROWS - 1) / MLP_ROWS), b(MLP_THREADS);
    const size_t lds = mlp_lds_bytes(A);
    hipError_t e = hipSuccess;
    // the opt-in for > 48 KB of dynamic LDS is per kernel and device and sticky: raise it only when a launch needs more
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
#define EB_MLP_LAUNCH(RT, CT)                                                                                     \
    do {                                                                                                          \
        static size_t granted[64];                                                                                \
        if (lds > 48 * 1024 && lds > granted[dev]) {                                                              \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_kernel<RT, CT>),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            if (e == hipSuccess) granted[dev] = lds;                                                              \
        }                                                                                                         \
        if (e == hipSuccess) hipLaunchKernelGGL((mlp_kernel<RT, CT>), g, b, lds, s, A);                           \
    } while (0)
    switch (A.units) {
        case 64: EB_MLP_LAUNCH(1, 1); break;
        case 128: EB_MLP_LAUNCH(2, 1); break;
        case 256: EB_MLP_LAUNCH(2, 2); break;
        default: EB_MLP_LAUNCH(2, 4); break;
    }
#undef EB_MLP_LAUNCH
    return e != hipSuccess ? e : hipGetLastError();
}

// Host side of eb_mlp_set_layer: Keras kernel [k_real, cols_real] row-major -> per 32-column tile, per step of
// 8 inputs, per lane, the 4 values that lane feeds to 4 consecutive MFMAs: W[2m + (lane >> 5)][tile*32 + (lane & 31)].
void pack_weights(const float* kernel, int k_real, int cols_real, int k_pad, int col_tiles, float* out) {
    const int steps = k_pad / 8;
    for (int ct = 0; ct < col_tiles; ++ct)
        for (int s = 0; s < steps; ++s)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q) {
                    const int k = 2 * (s * 4 + q) + (l >> 5), j = ct * 32 + (l & 31);
                    out[(((size_t)ct * steps + s) * 64 + l) * 4 + q] =
                        (k < k_real && j < cols_real) ? kernel[(size_t)k * cols_real + j] : 0.0f;
                }
}

// Host side of eb_mlp_set_layer for the OUTPUT layer (16 x 16 x 4 tiles): per 16-column tile, per group of four MFMA steps
// (16 inputs), per lane, the 4 values that lane feeds to those four MFMAs: W[4 (4 g + q) + (lane >> 4)][tile * 16 + (lane & 15)].
void pack_weights16(const float* kernel, int k_real, int cols_real, int k_pad, float* out) {
    const int groups = k_pad / 16, tiles = (cols_real + 15) / 16;
    for (int ct = 0; ct < tiles; ++ct)
        for (int g = 0; g < groups; ++g)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * (4 * g + q) + (l >> 4), j = ct * 16 + (l & 15);
                    out[(((size_t)ct * groups + g) * 64 + l) * 4 + q] = (k < k_real && j < cols_real) ? kernel[(size_t)k * cols_real + j] : 0.0f;
                }
}

// (the shield's punish accumulation — hier_decision.py:93-97 — rides on the rollout step's launch since round 6: eb_rollout.hip, env_wave)

}  // namespace eb
