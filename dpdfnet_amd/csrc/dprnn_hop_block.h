// dprnn_hop_block.h -- single-hop streaming: ONE launch per DPRNN block.
//
// A hop's DPRNN block was two dependent launches: the intra-band bi-GRU scan on 4-row tiles (gru_scan4.h: 2 x streams / 4
// workgroups, 48 dependent steps) and the glue (fcln_gi.h: one 16-row tile per workgroup, everything up to the next block's input
// projection).  The glue launch spent its first ~5 us on the launch ramp and on pulling 245 KB of operands per workgroup through
// its CU's L1 -- none of which depends on the scan.  Here both are workgroups of the same launch:
//   * blocks [0, 2 nx): the scans (block 2 x + dir; waves 4-7 leave at once).  h' goes out through agent-scope (write-through)
//     stores; behind the last step the workgroup drains its stores (s_waitcnt 0), meets at a barrier and publishes
//     flag[dir * nx + x] = epoch with one agent-scope store;
//   * blocks [2 nx, 2 nx + tiles): the glue tiles.  Each fetches its operands, residual rows and carried state, then polls the
//     flags of the scan workgroups its 16 rows come from (one or two per direction) and reads the rows with agent-scope loads.
// Forward progress: workgroups are dispatched in block order, so every scan workgroup is resident (or done) before any glue
// workgroup of its launch starts to wait, and a scan waits for nobody.  A glue tile whose flag does not arrive gives up through
// the GRU-256 clusters' time-out (cluster_spin_expired: device error flag, DPDF_E_RUNTIME at the next synchronisation point,
// stream calls recover from their snapshot).  The epoch is a per-branch launch counter (host side), so flags never need
// resetting between launches.  Hand-off price measured beforehand (tools/handoff_probe.hip (d)): 1.3-1.7 us for a flag behind
// 1 KB of fresh payload -- the rows of the last step; everything older has long arrived.
// Results are bit-identical to the two-launch form (same code, same order of operations).  Reference: onnx_model/layers.py:159-196.
#pragma once
#include "gru_scan4.h"
#include "fcln_gi.h"

struct HopBlockArgs {
    Gru64Args scan; const float* wfrag4; const float* gi_in; int gw;     // the scan half (gru64_scan4_gi_kernel's arguments)
    HopGlueArgs glue;                                                    // the glue half (dprnn_hop_glue8_kernel's)
    unsigned* flags; unsigned epoch; int nscan_x; int Fp; int* err;
    unsigned* done;                                                      // last block of a stack: see HopHandoff::done
};

template <bool NEXT>
__global__ __launch_bounds__(512) void dprnn_hop_block_kernel(HopBlockArgs b) {
    const int nscan = 2 * b.nscan_x;
    if ((int)blockIdx.x < nscan) {
        if (threadIdx.x >= 256) return;
        const int x = blockIdx.x >> 1, dir = blockIdx.x & 1;
        gru64_scan4_body<true>(b.scan, b.wfrag4, b.gi_in, b.gw, x, dir);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                 // every store of this wave has been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(b.flags + dir * b.nscan_x + x, b.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DPDF_PHASE_TRACE
        if (NEXT && b.Fp >= 48 && blockIdx.x == 0 && threadIdx.x == 0) dpdf_trace_buf[19] = __builtin_amdgcn_s_memtime();
#endif
    } else {
        dprnn_hop_glue8_body<NEXT, true>(b.glue, (int)blockIdx.x - nscan, HopHandoff{b.flags, b.epoch, b.nscan_x, b.Fp, b.err, b.done});
    }
}
