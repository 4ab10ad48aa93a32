// Measured and NOT shipped (docs/HISTORY.md section 7): kept out of the library build.  Was part of csrc/gru_scan.h up to round 2;
// include it behind gru_scan.h to rebuild the experiment (launch code: git history of dpdf_model.hip, run_gru256 / "gru256_pair").
// ---------------------------------------------------------------------------------------------
// gru256_ring_kernel<NT>: the 4-workgroup cluster scan with NT 16-row tiles per cluster, round-robin -- the form for
// launches that matter to THROUGHPUT (>= 8 tiles = 128 streams): what the pipeline pays for the GRU-256 scans is their
// chip footprint (CUs x time), and the one-tile form wastes half of it.  Measured on the one-tile kernel
// (tools/g256_instr.py, s_memtime): a 5.1 us step = 2.6 us of MFMAs + ~0.8 us of VALU (address arithmetic, AGPR<->VGPR
// copies of weights the register allocator parked in AccVGPRs: 375 VALU instructions per 192 MFMAs) + ~1.8 us waiting
// for the sweep's own agent-scope loads to come back.  Here:
//   * the cluster owns tiles 0..NT-1 (independent recurrences, the same W_hh slices) and runs blocks k = t*NT + e
//     round-robin; the sweep is split -- at the boundary after block k the loads that block k+2 needs are ISSUED
//     (their granules were published >= NT-2 blocks ago) and the loads issued one boundary earlier are CONSUMED for
//     block k+1 -- so the round trip of the loads lies under a whole MFMA block;
//   * W_hh lives in AccVGPRs on purpose and feeds the MFMAs directly from there (srcB = AGPR, inline asm: the
//     allocator otherwise copies every weight through a VGPR before its MFMA -- a first version of this kernel
//     spent 368 v_accvgpr moves per block);
//   * every global address is a wave-uniform base (scalar registers) + one lane offset.
// ~3 us per block instead of 5.1 on a QUARTER of the workgroups (NT = 4: 16 CUs for 256 clips, not 64).  Latency per
// step is NT blocks, so launches of few tiles keep the one-tile / eight-workgroup forms (run_gru256).  Granule protocol
// and buffer layout are those of gru256_cluster_kernel; a stale granule at consume time falls back to the spinning sweep.
// gi / out are addressed with UNCLAMPED rows: run_gru256 allocates them with the clip count rounded up to 16; rows >= B
// compute on whatever the padding holds (MFMA rows are independent) and are never stored.

// D(a[]) += A(v) * B(a): fp32 16x16x4 MFMA with the B operand taken from an AccVGPR
__device__ __forceinline__ void mfma16_accb(f32x4& c, float a, float b_acc) {
    // not volatile: ordered by the data dependence on c only, so the scheduler may move LDS reads / address math across
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ float to_acc(float v) {       // park a value in an AccVGPR
    float r;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
    return r;
}

template <int NT>
__global__ __launch_bounds__(256, 1) void gru256_ring_kernel(Gru256CArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[NT][2][16][260];    // [tile][buffer][row][unit]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave id, provably uniform
    const int cl = lane & 15, q = lane >> 4;
    const int ngroups = gridDim.x >> 2;
    int gp, j;
    if ((ngroups & 7) == 0) {      // keep a cluster on one XCD (block b -> XCD b % 8): speed only
        gp = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5);
        j = (blockIdx.x >> 3) & 3;
    } else {
        gp = blockIdx.x >> 2; j = blockIdx.x & 3;
    }
    const int ucol = 64 * j + 16 * w + cl;     // this lane's hidden unit
    const int tile0 = NT * gp;                 // tile e of the cluster covers rows [16 (tile0 + e), +16)

    float wr[64], wz[64], wn[64];              // AccVGPR-resident
    {
        const float* wf = a.whh_frag + ((size_t)(j * 4 + w) * 3) * 64 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            wr[k] = to_acc(wf[(size_t)(0 * 64 + k) * 64]);
            wz[k] = to_acc(wf[(size_t)(1 * 64 + k) * 64]);
            wn[k] = to_acc(wf[(size_t)(2 * 64 + k) * 64]);
        }
    }
    const float bhn = a.b_hn[ucol];
    // lane offsets (elements): one per kind, shared by all tiles and steps
    // (BYTE offsets: the scalar-base + 32-bit-vgpr-offset addressing mode wants zext(vgpr) added as bytes)
    const unsigned gi_lane = ((unsigned)(q * 4) * (unsigned)a.Tc * 768u + (unsigned)ucol) * 4u;
    const unsigned out_lane = ((unsigned)(q * 4) * (unsigned)a.Tc * 256u + (unsigned)ucol) * 4u;
    const unsigned pub_lane = ((unsigned)(q * 4) * 256u + (unsigned)ucol) * 8u;
    const unsigned gi_rs = (unsigned)a.Tc * 768u, out_rs = (unsigned)a.Tc * 256u;       // row strides in elements (uniform)

    float h[NT][4];
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int row0 = (tile0 + e) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + q * 4 + i;
            h[e][i] = r < a.B ? a.hstate[(long)r * a.h_stride + ucol] : 0.f;
        }
        for (int idx = tid; idx < 16 * 256; idx += 256) {
            const int r = idx >> 8, u = idx & 255;
            Hs[e][0][r][u] = row0 + r < a.B ? a.hstate[(long)(row0 + r) * a.h_stride + u] : 0.f;
        }
    }
    __syncthreads();

    // sweep geometry: granule k = 4 s + rr of this lane sits at row 4 rr + w, unit 64 ((j + 1 + s) & 3) + lane
    const unsigned lane_b8 = (unsigned)lane * 8u;      // unsigned 32-bit BYTE lane offsets: (scalar base) + zext(vgpr) addressing
    unsigned long long xv[12];     // the sweep in flight (issued at the previous block boundary)
    bool dead = false;             // a sweep timed out (or another workgroup's did): stop waiting, the host reports DPDF_E_RUNTIME
    auto sweep_issue = [&](int tile, int t) __attribute__((always_inline)) {
        const unsigned long long* slot = a.xbuf + ((size_t)(tile0 + tile) * 2 + (t & 1)) * 16 * 256;     // uniform
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const unsigned long long* p = slot + (4 * rr + w) * 256 + 64 * ((j + 1 + s) & 3);       // uniform
                xv[4 * s + rr] = __hip_atomic_load((const unsigned long long*)((const char*)p + lane_b8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    };
    auto sweep_finish = [&](int tile, int t, float (*Hn)[260]) __attribute__((always_inline)) {
        const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
        unsigned spins = 0;
        for (;;) {
            bool all_in = true;
#pragma unroll
            for (int k = 0; k < 12; ++k) all_in &= (unsigned)(xv[k] >> 32) == epoch;
            if (__builtin_expect(all_in, 1)) break;
            if (dead || cluster_spin_expired(spins, a.err, dead)) break;
            __builtin_amdgcn_s_sleep(1);
            sweep_issue(tile, t);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Hn[4 * rr + w][64 * ((j + 1 + s) & 3) + lane] = __uint_as_float((unsigned)xv[4 * s + rr]);
    };

    for (int t = 0; t < a.Tc; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            const int row0 = (tile0 + e) * 16;
            // ---- block: step t of tile e ----
            float gr[4], gz[4], gn[4];
            {   // this step's input projections: needed only after the MFMA loop, their latency hides under it
                const float* g = a.gi + ((size_t)row0 * a.Tc + t) * 768;       // uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* gp_ = (const char*)(g + (size_t)i * gi_rs);      // uniform
                    gr[i] = *(const float*)(gp_ + gi_lane); gz[i] = *(const float*)(gp_ + 1024 + gi_lane); gn[i] = *(const float*)(gp_ + 2048 + gi_lane);
                }
            }
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, ahn = {bhn, bhn, bhn, bhn};
            const float* hrow = &Hs[e][cur][cl][4 * q];
            float4 h4[16];                               // the whole A panel of the step up front (64 VGPRs)
#pragma unroll
            for (int c = 0; c < 16; ++c) h4[c] = *(const float4*)(hrow + 16 * c);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float hv[4] = {h4[c].x, h4[c].y, h4[c].z, h4[c].w};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    mfma16_accb(ar, hv[kb], wr[c * 4 + kb]);
                    mfma16_accb(az, hv[kb], wz[c * 4 + kb]);
                    mfma16_accb(ahn, hv[kb], wn[c * 4 + kb]);
                }
            }
            // the MFMAs are opaque to the compiler's hazard recogniser: XDL write -> VALU read of the accumulators
            // needs 11 wait states after an 8-pass MFMA (CDNA3 ISA, manually inserted wait states); 24 given.  The
            // accumulators are operands so that the nops sit between the last MFMA and the first read.
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(ar), "+a"(az), "+a"(ahn));
            const unsigned epoch = a.epoch_base + (unsigned)t + 1u;
            unsigned long long* slot = a.xbuf + ((size_t)(tile0 + e) * 2 + (t & 1)) * 16 * 256;        // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = sigmoid_f(ar[i] + gr[i]);
                const float z = sigmoid_f(az[i] + gz[i]);
                const float n = gru_candidate(r, ahn[i], gn[i]);
                h[e][i] = gru_blend(z, n, h[e][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __hip_atomic_store((unsigned long long*)((char*)(slot + i * 256) + pub_lane),
                                   ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(h[e][i]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float* o = a.out + ((size_t)row0 * a.Tc + t) * 256;                // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Hs[e][nxt][q * 4 + i][ucol] = h[e][i];
                if (row0 + q * 4 + i < a.B) *(float*)((char*)(o + (size_t)i * out_rs) + out_lane) = h[e][i];
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- boundary ----
            // consume the loads issued one boundary ago: the peers' slices the NEXT block needs
            // (tile e+1 at step t needs h'(t-1); after the last tile, tile 0 at step t+1 needs h'(t))
            const int n1 = (e + 1) % NT;
            const int t1 = e + 1 < NT ? t - 1 : t;
            if (t1 >= 0 && (e + 1 < NT || t + 1 < a.Tc)) sweep_finish(n1, t1, Hs[n1][(t1 + 1) & 1]);
            // issue the loads for the block after that (tile e+2, same rule): published >= NT-2 blocks ago
            const int n2 = (e + 2) % NT;
            const int t2 = e + 2 < NT ? t - 1 : t;
            if (t2 >= 0 && (e + 2 < NT || t + 1 < a.Tc)) sweep_issue(n2, t2);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    // every tile's LAST step, synchronously: the guarantee that each peer has consumed the carried state (it has
    // published its last step) before this workgroup overwrites its slice of it
    if (a.Tc > 0) {
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            sweep_issue(e, a.Tc - 1);
            sweep_finish(e, a.Tc - 1, Hs[e][a.Tc & 1]);
        }
    }
#pragma unroll
    for (int e = 0; e < NT; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tile0 + e) * 16 + q * 4 + i;
            if (r < a.B) a.hstate[(long)r * a.h_stride + ucol] = h[e][i];
        }
}

