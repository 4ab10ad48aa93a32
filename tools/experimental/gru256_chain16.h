// Measured and NOT shipped (docs/HISTORY.md section 7): kept out of the library build.  Was part of csrc/gru_stack.h up to round 2
// (launch code: git history of dpdf_model.hip, run_gru256_chain / "gru256_chain").
// ---------------------------------------------------------------------------------------------
// gru256_chain16_kernel: ALL FIVE GRUCell(256) layers of stage 2 as one wavefront launch (<= 2 tiles).
//   role 0  E   embedding cell (reference onnx_model/dpdfnet.py:233-241): hoisted input projection, as cell A above
//   role 1/2    ERB decoder cells 0, 1 (dpdfnet.py:343-360)      role 3/4    DF decoder cells 0, 1 (dpdfnet.py:486-500)
// Between the embedding cell and the decoders' first cells sit two grouped linears with ReLU (emb = relu(GL_out(h_E)),
// 16 groups 16 -> 32; x = relu(GL_in(emb)), ERB decoder 16 groups 32 -> 16, DF decoder 8 groups 64 -> 32).  Grouped =
// block-diagonal: the 16 input units of decoder slice j depend on h_E units [16 j, 16 j + 16) only (ERB) or on
// [32 (j/2), 32 (j/2) + 32) (DF).  So embedding workgroup j computes both slices j itself -- 160 FMAs per thread on the
// VALU, for the PREVIOUS frame, under the round trip of its own granule sweep -- and publishes them into two per-frame
// rings; the decoders' first cells then run exactly like cell B above (input partials W_ih x from the upstream ring, one
// step behind, recurrent partials on top).  The chain of dependent scan steps of a chunk is T + 5 instead of 3 T.
// Every cell publishes h(t) into its own per-frame ring [tile][Tc][16][256] (peers sweep slot t, the downstream cell
// prefetches slot t + 3 a step ahead).  80 workgroups per tile, all co-resident.
struct Gru256ChainCell {
    const float* whh; const float* wih; const float* bias; const float* bhn;   // wih (packed like whh) / bias [768]: followers only
    float* hstate; float* out;
    unsigned long long* ring;          // own h per frame
    const unsigned long long* up;      // followers: upstream ring (x per frame)
};
struct Gru256ChainArgs {
    Gru256ChainCell c[5];
    const float* gi;                   // E: hoisted input projection [B*Tc][768]
    const float* w_out; const float* b_out;      // GL_out  [16][32][16], [512]
    const float* w_ine; const float* b_ine;      // ERB GL_in [16][16][32], [256]
    const float* w_ind; const float* b_ind;      // DF  GL_in [8][32][64],  [256]
    unsigned long long* xr_e; unsigned long long* xr_d;     // x rings of the two decoders' first cells
    long h_stride; int B, Tc; unsigned epoch_base; int* err;
};

__global__ __launch_bounds__(256, 1) void gru256_chain16_kernel(Gru256ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[2][16][260];
    __shared__ __attribute__((aligned(16))) float Ha[2][16][260];      // followers: upstream x of frames t+1 / t+2; E: GL scratch
    __shared__ float Ps[4][4][4][64];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int cl = lane & 15, q = lane >> 4;
    const int rt = blockIdx.x / 80, rem = blockIdx.x - rt * 80, role = rem >> 4, j = rem & 15;
    const bool fol = role != 0;
    const Gru256ChainCell& C = a.c[role];
    const int row0 = rt * 16, u0 = 16 * j, Tc = a.Tc;

    float wr[16], wz[16], wn[16], xr_[16], xz_[16], xn_[16];
    {
        const float* wf = C.whh + ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) { wr[k] = wf[(size_t)(0 * 64 + k) * 64]; wz[k] = wf[(size_t)(1 * 64 + k) * 64]; wn[k] = wf[(size_t)(2 * 64 + k) * 64]; }
        const float* xf = (fol ? C.wih : C.whh) + ((size_t)j * 3) * 64 * 64 + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            xr_[k] = fol ? xf[(size_t)(0 * 64 + k) * 64] : 0.f; xz_[k] = fol ? xf[(size_t)(1 * 64 + k) * 64] : 0.f; xn_[k] = fol ? xf[(size_t)(2 * 64 + k) * 64] : 0.f;
        }
    }
    const float bhn = C.bhn[u0 + cl];
    const float b_r = fol ? C.bias[u0 + cl] : 0.f, b_z = fol ? C.bias[256 + u0 + cl] : 0.f, b_n = fol ? C.bias[512 + u0 + cl] : 0.f;
    const int r_own = row0 + q * 4 + w;
    const bool ok = r_own < a.B;
    const int rc = ok ? r_own : a.B - 1;
    float h_own = C.hstate[(long)rc * a.h_stride + u0 + cl];
    for (int idx = tid; idx < 16 * 256; idx += 256) {
        int r = idx >> 8, u = idx & 255;
        int rr = row0 + r < a.B ? row0 + r : a.B - 1;
        Hs[0][r][u] = C.hstate[(long)rr * a.h_stride + u];
    }
    const size_t ring_off = (size_t)rt * Tc * 16 * 256;
    unsigned long long* ring = C.ring + ring_off;
    const unsigned long long* up = fol ? C.up + ring_off : nullptr;
    bool dead = false;
    const int sr_ = tid >> 4, su_ = tid & 15;

    // ---- E: grouped-linear blocks of this workgroup in LDS (inside Ha: E has no upstream), transposed so that lanes read
    // consecutive words: Wo[half 2][k 16][n 32] | Wie[k 32][c 16] | Wid[k 64][c 16] | E64 [16][65]
    float* gl = &Ha[0][0][0];
    float* Wo = gl; float* Wie = gl + 1024; float* Wid = gl + 1536; float* E64 = gl + 2560;
    const int gp = j >> 1, hs = j & 1;
    float bo[2][2] = {{0.f, 0.f}, {0.f, 0.f}}; float bie = 0.f, bid = 0.f;
    if (!fol) {
        for (int i = tid; i < 1024; i += 256) { const int h = i >> 9, k = (i >> 5) & 15, n = i & 31; Wo[i] = a.w_out[((size_t)(2 * gp + h) * 32 + n) * 16 + k]; }
        for (int i = tid; i < 512; i += 256) { const int k = i >> 4, c = i & 15; Wie[i] = a.w_ine[((size_t)j * 16 + c) * 32 + k]; }
        for (int i = tid; i < 1024; i += 256) { const int k = i >> 4, c = i & 15; Wid[i] = a.w_ind[((size_t)gp * 32 + 16 * hs + c) * 64 + k]; }
        bo[0][0] = a.b_out[(2 * gp) * 32 + su_]; bo[0][1] = a.b_out[(2 * gp) * 32 + 16 + su_];
        bo[1][0] = a.b_out[(2 * gp + 1) * 32 + su_]; bo[1][1] = a.b_out[(2 * gp + 1) * 32 + 16 + su_];
        bie = a.b_ine[16 * j + su_]; bid = a.b_ind[32 * gp + 16 * hs + su_];
    }
    // E: x slices of frame f for both decoders from h_E(f) in Hs[buf]; two phases around a workgroup barrier
    auto gl_phase1 = [&](int buf) {
        float e[2][2] = {{bo[0][0], bo[0][1]}, {bo[1][0], bo[1][1]}};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float hv = Hs[buf][sr_][32 * gp + 16 * h + k];
                e[h][0] = __builtin_fmaf(hv, Wo[(h * 16 + k) * 32 + su_], e[h][0]);
                e[h][1] = __builtin_fmaf(hv, Wo[(h * 16 + k) * 32 + 16 + su_], e[h][1]);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) { E64[sr_ * 65 + 32 * h + su_] = fmaxf(e[h][0], 0.f); E64[sr_ * 65 + 32 * h + 16 + su_] = fmaxf(e[h][1], 0.f); }
    };
    auto gl_phase2 = [&](int f) {
        float xe = bie, xd = bid;
#pragma unroll
        for (int k = 0; k < 32; ++k) xe = __builtin_fmaf(E64[sr_ * 65 + 32 * hs + k], Wie[k * 16 + su_], xe);
#pragma unroll
        for (int k = 0; k < 64; ++k) xd = __builtin_fmaf(E64[sr_ * 65 + k], Wid[k * 16 + su_], xd);
        const unsigned long long ep = (unsigned long long)(a.epoch_base + (unsigned)f + 1u) << 32;
        const size_t o = ring_off + (size_t)f * 16 * 256 + sr_ * 256 + u0 + su_;
        __hip_atomic_store(a.xr_e + o, ep | (unsigned long long)__float_as_uint(fmaxf(xe, 0.f)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.xr_d + o, ep | (unsigned long long)__float_as_uint(fmaxf(xd, 0.f)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    auto fetch_up = [&](int f) {
        const unsigned ep = a.epoch_base + (unsigned)f + 1u;
        const unsigned long long* slot = up + (size_t)f * 16 * 256;
        unsigned long long xv[16];
        unsigned spins = 0;
        for (;;) {
            bool all_in = true;
#pragma unroll
            for (int k = 0; k < 16; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 16; ++k) all_in &= (unsigned)(xv[k] >> 32) == ep;
            if (all_in) break;
            if (dead || cluster_spin_expired(spins, a.err, dead)) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) Ha[f & 1][sr_][16 * k + su_] = __uint_as_float((unsigned)xv[k]);
    };
    f32x4 pi_r = {0.f, 0.f, 0.f, 0.f}, pi_z = pi_r, pi_n = pi_r;
    auto ih_part = [&](int f) {
        pi_r = f32x4{0.f, 0.f, 0.f, 0.f}; pi_z = pi_r; pi_n = pi_r;
        const float* arow = &Ha[f & 1][cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h4 = *(const float4*)(arow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pi_r = mfma16(hv[kb], xr_[c * 4 + kb], pi_r);
                pi_z = mfma16(hv[kb], xz_[c * 4 + kb], pi_z);
                pi_n = mfma16(hv[kb], xn_[c * 4 + kb], pi_n);
            }
        }
    };

    float gr = 0.f, gz = 0.f, gn = 0.f;
    unsigned long long av[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) av[k] = 0ull;
    if (!fol) {
        const float* g = a.gi + ((size_t)rc * Tc) * 768 + u0 + cl;
        gr = g[0]; gz = g[256]; gn = g[512];
    } else {
        fetch_up(0);
        if (Tc > 1) fetch_up(1);
        if (Tc > 2) {
            const unsigned long long* slot_a = up + (size_t)2 * 16 * 256;
#pragma unroll
            for (int k = 0; k < 16; ++k) av[k] = __hip_atomic_load(slot_a + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (fol) ih_part(0);

    int cur = 0;
    for (int t = 0; t < Tc; ++t) {
        f32x4 pr = pi_r, pz = pi_z, pn = {0.f, 0.f, 0.f, 0.f};
        const float* hrow = &Hs[cur][cl][64 * w + 4 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 h4 = *(const float4*)(hrow + 16 * c);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                pr = mfma16(hv[kb], wr[c * 4 + kb], pr);
                pz = mfma16(hv[kb], wz[c * 4 + kb], pz);
                pn = mfma16(hv[kb], wn[c * 4 + kb], pn);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { Ps[w][0][i][lane] = pr[i]; Ps[w][1][i][lane] = pz[i]; Ps[w][2][i][lane] = pi_n[i]; Ps[w][3][i][lane] = pn[i]; }
        const float xr = fol ? b_r : gr, xz = fol ? b_z : gz, xn = fol ? b_n : gn;
        if (!fol && t + 1 < Tc) {
            const float* g = a.gi + ((size_t)rc * Tc + t + 1) * 768 + u0 + cl;
            gr = g[0]; gz = g[256]; gn = g[512];
        }
        __syncthreads();
        const int nxt = cur ^ 1;
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned long long* slot = ring + (size_t)t * 16 * 256;
        {
            float sr = Ps[0][0][w][lane], sz = Ps[0][1][w][lane], sx = Ps[0][2][w][lane], sn = Ps[0][3][w][lane];
#pragma unroll
            for (int k = 1; k < 4; ++k) { sr += Ps[k][0][w][lane]; sz += Ps[k][1][w][lane]; sx += Ps[k][2][w][lane]; sn += Ps[k][3][w][lane]; }
            const float r = sigmoid_f(xr + sr);
            const float z = sigmoid_f(xz + sz);
            const float n = gru_candidate(r, bhn + sn, xn + sx);
            h_own = gru_blend(z, n, h_own);
        }
        __hip_atomic_store(slot + (q * 4 + w) * 256 + u0 + cl,
                           ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h_own),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Hs[nxt][q * 4 + w][u0 + cl] = h_own;
        if (ok && C.out) C.out[((size_t)rc * Tc + t) * 256 + u0 + cl] = h_own;
        {
            unsigned long long xv[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * ((j + 1 + k) & 15) + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
            if (fol) { if (t + 1 < Tc) ih_part(t + 1); }
            else if (t > 0) {                 // E: decoder inputs of frame t-1 (h_E(t-1) is Hs[cur], complete since the last barrier)
                gl_phase1(cur);
                __syncthreads();
                gl_phase2(t - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned spins = 0;
            for (;;) {
                bool all_in = true;
#pragma unroll
                for (int k = 0; k < 15; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
                if (all_in) break;
                if (dead || cluster_spin_expired(spins, a.err, dead)) break;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < 15; ++k) xv[k] = __hip_atomic_load(slot + sr_ * 256 + 16 * ((j + 1 + k) & 15) + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) Hs[nxt][sr_][16 * ((j + 1 + k) & 15) + su_] = __uint_as_float((unsigned)xv[k]);
        }
        if (fol) {
            if (t + 2 < Tc) {
                const unsigned ep_a = a.epoch_base + (unsigned)t + 3u;
                bool a_in = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) a_in &= (unsigned)(av[k] >> 32) == ep_a;
                if (a_in) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) Ha[t & 1][sr_][16 * k + su_] = __uint_as_float((unsigned)av[k]);
                } else {
                    fetch_up(t + 2);
                }
            }
            if (t + 3 < Tc) {
                const unsigned long long* slot_a = up + (size_t)(t + 3) * 16 * 256;
#pragma unroll
                for (int k = 0; k < 16; ++k) av[k] = __hip_atomic_load(slot_a + sr_ * 256 + 16 * k + su_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (!fol) {            // decoder inputs of the last frame
        gl_phase1(cur);
        __syncthreads();
        gl_phase2(Tc - 1);
    }
    if (ok) C.hstate[(long)rc * a.h_stride + u0 + cl] = h_own;
}

