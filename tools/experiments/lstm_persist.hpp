// lstm_persist.hpp — the whole recurrence of one bidirectional LSTM layer in ONE launch
// (torch.nn.LSTM semantics as in lstm.hpp; the reference runs it inside its opaque TorchScript model,
// pero_ocr/ocr_engine/pytorch_ocr_engine.py:66-69; aten::lstm / mkldnn_rnn_layer in the CPU profile).
//
// Why: the per-step kernel of lstm.hpp needs 2*T*L dependent launches (288 per 256-line chunk) that re-read
// W_hh from L2 every step and interleave with the next chunk's MFMA-bound conv kernels.  The recurrence is only 2 %
// of a chunk's FLOPs, but strictly serial in t.  What is NOT serial: lines.  So the layer is laid out as a
// wavefront over (step, 16-line slice):
//
//   grid = (H/32 unit groups, 2 directions[, slice interleave]) workgroups of 8 waves - 16 workgroups for H = 256 -
//   resident for the whole layer.  Wave w of unit group p owns 4 hidden units x 4 gates = 16 gate columns and keeps that slice of
//   W_hh in REGISTERS for all T steps (H/16 float4 per lane = 64 VGPRs at H = 256): W_hh is read once per layer.
//   Every workgroup walks the same sequence of (step, slice) items.  Item (t, s) needs h_{t-1} of the 16 lines of
//   slice s from ALL unit groups of its direction - published a whole sweep over the other slices earlier, so the
//   hand-off latency hides behind the other slices' work (with >= 3 slices nobody normally waits).
//
//   h hand-off: h_t IS the layer output row y[row(line, t)][dir*H + unit], written once per launch with
//   write-through (sc1) stores; every wave counts its finished items on a per-(direction, slice) counter
//   (agent-scope atomic, after its stores have drained); a consumer polls that one word (relaxed, agent scope),
//   then reads the rows with sc1 loads (L2-served: a CU's L1 is never refreshed by other CUs' stores).
//   Protocol = cdna_hip_programming.md Guideline 16 (R1), placement-independent; every spin is bounded.
//   The inputs of the NEXT item are requested before the MFMAs of the current one (when already published), and a
//   wave publishes item i only just before it stores item i+1, when its stores of item i have long drained - so
//   neither the hand-off read nor the write-through latency sits on the item's critical path.
//
//   Per item and wave: the h slice [16 lines][H] is loaded straight into MFMA A-operand registers (16 rows x 64 B per
//   load instruction); 4*H/16 v_mfma_f32_16x16x4_f32 accumulate W_hh h on top of the hoisted input projection
//   (accumulator initialised from xproj); the four gates of a (line, unit) are gathered through a 1 KB per-wave LDS
//   transpose; sigmoid/tanh cell update with c kept in HBM (L2-resident, private to the lane that owns it).
//   Waves never synchronise with each other inside the launch (no barrier, no shared staging): a workgroup is just
//   8 waves = 2 per SIMD whose MFMA, VALU and memory phases overlap (measured: a lone wave per SIMD spends only a
//   third of an item's ~9000 cycles issuing MFMAs).
//
// Ragged launches: line i has line_T[i] frames; the backward direction starts at the line's own last frame;
// a slice is walked for slice_T[s] = max line_T of its lines steps; finished lines load zeros and store nothing.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hpp"
#include "lstm.hpp"

namespace pocr {

struct LstmPersistArgs {
    const float *xproj;     // [rows][8H]   (dir, gate, unit): W_ih x + b_ih + b_hh for every frame
    const float *whh_p;     // [2][H/4][H/16][64][4]: wave-private B fragments (build_whh_persist in pocr_hip.hip)
    float *y;               // [rows][2H]   layer output = the h hand-off buffer
    float *c;               // [2][npad][H] cell state scratch (needs no initialisation)
    unsigned *flags;        // [2][n_slices] finished-item counters, zeroed before the launch
    unsigned *err;          // set to 1 when a bounded spin gave up (zeroed before the launch)
    const int32_t *line_T, *row_off, *slice_T;
    int32_t n, npad, n_slices, T;
    int32_t y_bytes;        // size of y in bytes (< 2^31: buffer descriptor range; out-of-range loads return 0)
    int32_t dbg_mask;       // timing experiments only (results become wrong): 1 no h loads, 2 no xproj loads, 4 no c load, 8 no probe, 16 no stores
    unsigned long long *dbg;    // optional (POCR_LSTM_DBG=1): [items, blocking items, cycles: wait, publish, barrier, mfma, cell, total]
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int LSTM_PERSIST_MAX_LINES = 2048;      // per-line tables are kept in LDS (16.5 KB); larger launches use the step kernel

__device__ __forceinline__ bool lstm_wait_flag(unsigned *flag, unsigned need, unsigned *err) {
    unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= need) return true;
    const unsigned long long t0 = wall_clock64();           // 100 MHz
    while (true) {
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= need) return true;
        if (wall_clock64() - t0 > 300000000ull) {           // 3 s: a partner workgroup never became resident
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

__device__ __forceinline__ void lstm_publish(unsigned *flag, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have reached the fabric
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Registers of the item in flight.
template <int KGT>
struct LstmItemRegs {
    f32x4 a[KGT];           // h_{t-1} of the slice as MFMA A operands: lane (line, kq) holds h[line][16 kg + 4 kq + 0..3]
    f32x4 xp;               // xproj in MFMA D layout = accumulator initialisation
    float c;                // previous cell state of the lane's (line, unit)
    int crow;               // row (frame) the lane's cell writes, -1 = line not live at this step
    int xlive;              // bit j: xp[j] belongs to a live line (else it is a dummy load and counts as 0)
    int st, sx;             // the item
    bool have, fetched;     // item exists / its loads have been issued
};

// KGT = H / 16 (k-groups of 16): the loops over K unroll and the W_hh slice lives in registers.  NW waves per workgroup.
template <int KGT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void lstm_persist_kernel(LstmPersistArgs a) {
    constexpr int H = 16 * KGT;
    __shared__ __attribute__((aligned(16))) float trans[NW][256];         // per wave: [line 16][unit 4][gate 4]
    // per-line geometry in LDS: every item looks up the frame count and first row of 6 lines per lane, and through
    // global memory those dependent look-ups (index -> row -> address -> data) cost more than the item's MFMAs
    __shared__ int2 s_line[LSTM_PERSIST_MAX_LINES];               // {frames, first row} of every line (frames 0 beyond n)
    __shared__ int s_sliceT[LSTM_PERSIST_MAX_LINES / 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < a.npad; i += NW * 64) s_line[i] = i < a.n ? make_int2(a.line_T[i], a.row_off[i]) : make_int2(0, 0);
    for (int i = tid; i < a.n_slices; i += NW * 64) s_sliceT[i] = a.slice_T[i];
    __syncthreads();                                          // the only barrier of the launch
    const int dir = blockIdx.y;
    const int sx0 = blockIdx.z, sxs = gridDim.z;              // this workgroup's slices: sx0, sx0 + sxs, ...
    const int col = lane & 15, q = lane >> 4;                 // MFMA: A row (line) / B, D column; k quarter / D row quad
    const int gate = col >> 2, usub = col & 3;
    const int u0 = (blockIdx.x * NW + wave) * 4;              // first of this wave's 4 hidden units
    const unsigned per_item = H / 4;                          // waves per direction = counter increments per item
    const int cl = lane >> 2, cu = lane & 3;                  // cell role (after the transpose): line, unit

    // ---- W_hh slice -> registers: B operand of k-step (kg, j), lane (col, q) = W_hh[gate*H + u0 + usub][16 kg + 4 q + j]
    f32x4 w[KGT];
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.whh_p) + ((size_t)(dir * (H / 4) + u0 / 4) * KGT) * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < KGT; ++kg) w[kg] = wp[(size_t)kg * 64];
    }
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    unsigned *flags = a.flags + (size_t)dir * a.n_slices;

    // next (step, slice) item after (st, sx), step-major over this workgroup's slices that still run; false at the end
    auto advance = [&](int &st, int &sx) -> bool {
        while (st < a.T) {
            sx = sx < 0 ? sx0 : sx + sxs;
            if (sx >= a.n_slices) { sx = -1; ++st; continue; }
            if (st < s_sliceT[sx]) return true;
        }
        return false;
    };

    // issues every load of the item in R; the caller has made sure that the item's h rows are published.  Branch-free:
    // the per-line look-ups come from LDS in one go, loads of finished lines go to a harmless address (row 0 / beyond
    // the buffer range, which returns 0) and are masked when consumed.
    auto fetch = [&](LstmItemRegs<KGT> &R) {
        const int st = R.st, sx = R.sx;
        const int2 ls = s_line[sx * 16 + col];                // A operand: line = lane & 15 (= col), kq = q
        int2 lx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lx[j] = s_line[sx * 16 + 4 * q + j];      // xproj rows in D layout: reg j = line 4q + j
        const int2 lc = s_line[sx * 16 + cl];                 // cell role
        {
            const bool live = st > 0 && st < ls.x;
            const int tp = dir == 0 ? st - 1 : ls.x - st;     // the frame this line processed one step earlier
            const int rowb = live ? ((ls.y + tp) * (2 * H) + dir * H + q * 4) * 4 : 0x7ffff000;    // out of range: loads return 0
#pragma unroll
            for (int kg = 0; kg < KGT; ++kg)
                if (!(a.dbg_mask & 1)) R.a[kg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, rowb + kg * 64, 0, 16));
        }
        const float *xcol = a.xproj + dir * 4 * H + gate * H + u0 + usub;
        R.xlive = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = st < lx[j].x;
            const int t = dir == 0 ? st : lx[j].x - 1 - st;
            if (!(a.dbg_mask & 2)) R.xp[j] = xcol[(size_t)(live ? lx[j].y + t : 0) * (8 * H)];
            R.xlive |= live ? 1 << j : 0;
        }
        {
            const bool live = st < lc.x;
            const int t = dir == 0 ? st : lc.x - 1 - st;
            R.crow = live ? lc.y + t : -1;
            if (!(a.dbg_mask & 4)) R.c = a.c[((size_t)dir * a.npad + sx * 16 + cl) * H + u0 + cu];      // garbage at step 0: masked below
        }
        R.fetched = true;
    };

    LstmItemRegs<KGT> X;                         // the item being worked on; re-used for the next one once its MFMAs are issued
    unsigned *pend = nullptr;                   // counter of the item whose stores are still to be published
    int tst = 0, tsx = -1;
    X.have = advance(tst, tsx); X.st = tst; X.sx = tsx; X.fetched = false;
    if (!X.have) return;
    fetch(X);                                   // the first item is a step 0: nothing to wait for
    // `probe` = the item after X; its counter is read one item before it is looked at (a stale value of a monotonic
    // counter only delays the prefetch)
    int pst = tst, psx = tsx;
    bool phave = advance(pst, psx);
    unsigned pval = 0;
    if (phave && pst > 0) pval = __hip_atomic_load(flags + psx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    unsigned long long d_items = 0, d_block = 0, d_wait = 0, d_pub = 0, d_mfma = 0, d_cell = 0, d_store = 0, d_fetch = 0, d_first = 0, d_t40 = 0;
    const bool dbg = a.dbg != nullptr;
    const unsigned long long d_t0 = dbg ? clock64() : 0;
    while (X.have) {
        unsigned long long c0 = dbg ? clock64() : 0, c1;
        if (dbg && d_items == 40) d_t40 = c0 - d_t0;
        ++d_items;
        if (!X.fetched) {
            ++d_block;
            // its inputs were not published when we looked: publish everything we hold (others may be waiting for it -
            // with fewer than ~5 slices per workgroup they always are), then wait
            if (pend) { lstm_publish(pend, lane); pend = nullptr; }
            if (!lstm_wait_flag(flags + X.sx, (unsigned)X.st * per_item, a.err)) return;
            fetch(X);
        }
        if (dbg) { c1 = clock64(); d_wait += c1 - c0; c0 = c1; }
        f32x4 acc0, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc0[j] = (X.xlive >> j) & 1 ? X.xp[j] : 0.f;
        // ---- gates = xproj + W_hh h  (two accumulators: consecutive MFMAs are independent).  The wave in its MFMA phase
        //      outranks its SIMD partner (which is then in its VALU / memory phase and needs only the issue slots between
        //      MFMAs): without this the older wave of a SIMD always wins and the two drift apart by the full slack of the
        //      slice wavefront, after which every hand-off is a blocking one.
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int kg = 0; kg < KGT; ++kg) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.a[kg][0], w[kg][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.a[kg][1], w[kg][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.a[kg][2], w[kg][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(X.a[kg][3], w[kg][3], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (dbg) { c1 = clock64(); d_mfma += c1 - c0; c0 = c1; }
        const int crow = X.crow, cline = X.sx * 16 + cl, cst = X.st;
        const float cprev = X.c;
        unsigned *mine = flags + X.sx;
        // ---- publish the previous item: its stores were issued a whole MFMA phase ago and nothing younger is outstanding
        if (pend) lstm_publish(pend, lane);
        pend = nullptr;
        if (dbg) { c1 = clock64(); d_pub += c1 - c0; c0 = c1; }
        // ---- the operand registers are free: request the next item now if its inputs are published; its loads land while
        //      this item's gates are evaluated (and while the SIMD's other wave issues its MFMAs)
        X.have = phave; X.st = pst; X.sx = psx; X.fetched = false;
        if (phave) {
            const unsigned pv = __builtin_amdgcn_readfirstlane(pval);
            if ((a.dbg_mask & 8) || pv >= (unsigned)pst * per_item) fetch(X);
            else if (dbg && d_first == 0) d_first = ((unsigned long long)pst << 48) | ((unsigned long long)psx << 32) | pv;
            phave = advance(pst, psx);
            if (phave && pst > 0 && !(a.dbg_mask & 8)) pval = __hip_atomic_load(flags + psx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else pval = 0;
        }
        if (dbg) { c1 = clock64(); d_fetch += c1 - c0; c0 = c1; }
        // ---- D layout (lane = (col = gate*4 + unit, q), reg j = line 4q + j) -> cell layout (lane = line*4 + unit)
        float *tw = trans[wave];
#pragma unroll
        for (int j = 0; j < 4; ++j) tw[((4 * q + j) * 4 + usub) * 4 + gate] = acc0[j] + acc1[j];
        __builtin_amdgcn_wave_barrier();
        const f32x4 g4 = *reinterpret_cast<const f32x4 *>(&tw[lane * 4]);
        __builtin_amdgcn_wave_barrier();
        if (crow >= 0) {
            const float gi = sigmoid_f32(g4[0]);
            const float gf = sigmoid_f32(g4[1]);
            const float gg = tanhf(g4[2]);
            const float go = sigmoid_f32(g4[3]);
            const float cn = gf * (cst > 0 ? cprev : 0.f) + gi * gg;
            const float hn = go * tanhf(cn);
            if (!(a.dbg_mask & 16)) {
                a.c[((size_t)dir * a.npad + cline) * H + u0 + cu] = cn;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), yrs, (crow * (2 * H) + dir * H + u0 + cu) * 4, 0, 16);
            }
        }
        pend = mine;
        if (dbg) { c1 = clock64(); d_cell += c1 - c0; c0 = c1; }
    }
    if (pend) lstm_publish(pend, lane);
    if (dbg && tid == 0 && blockIdx.x == 0 && dir == 0 && sx0 == 0) {
        a.dbg[0] = d_items; a.dbg[1] = d_block; a.dbg[2] = d_wait; a.dbg[3] = d_pub; a.dbg[4] = d_first; a.dbg[5] = d_mfma;
        a.dbg[6] = d_cell; a.dbg[7] = clock64() - d_t0; a.dbg[8] = d_store; a.dbg[9] = d_fetch;
    }
}

}  // namespace pocr
