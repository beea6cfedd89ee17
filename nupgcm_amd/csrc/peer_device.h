// Device-side pieces of the peer-window transport (comm.hip): system-scope accessors, the bounded spin, the device view of a
// halo plan and the two halves of its consumer side.  (A header of its own since round 3 tried to let the Arnoldi kernel
// consume the halo in place - boundary tiles waiting for the flags and reading the window directly: no gain at one rank,
// +37 us per iteration on the one-GPU kernel from the second gather base at its 80-VGPR cap, and 768 spinning workgroups can
// starve a peer that shares the device)
#pragma once
#include "common.h"

namespace npg {

constexpr int kHaloWG = 64;        // most sender workgroups (= flags) per (sender, receiver) pair: one wave polls a peer's flags at once

// system-scope (cross-device) accesses: write-through stores / cache-bypassing loads (sc0 sc1 on gfx950)
// (global address space spelled out: global_load / global_store, never flat_)
typedef __attribute__((address_space(1))) uint64_t gu64;
__device__ __forceinline__ void st_sys(uint64_t *p, uint64_t v) {
    __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t ld_sys(const uint64_t *p) {
    return __hip_atomic_load((gu64 *)const_cast<uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys_f64(double *p, double v) {
    __hip_atomic_store((gu64 *)reinterpret_cast<uint64_t *>(p), (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// every wait is bounded: `ticks` of the 100 MHz constant clock, then the status word (pinned host memory) is set and the
// kernel carries on with whatever it has - the host sees the status at its next look and fails the call
struct SpinGuard {
    unsigned long long t0, ticks;
    int *status;
    int code;
    __device__ __forceinline__ SpinGuard(unsigned long long tk, int *st, int c) : t0(__builtin_amdgcn_s_memrealtime()), ticks(tk), status(st), code(c) {}
    __device__ __forceinline__ bool expired() {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 < ticks) return false;
        __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
};

// halo plan, device view (tables live in device memory; pointers into peers' windows are IPC mappings)
struct HaloPeerDev {
    // sender side: workgroup b gathers x[send_idx[s]] for s in [tab[b].x, tab[b].y) into peer tab[b].z's window and raises
    // that peer's flag tab[b].w
    const int4 *tab;
    const int64_t *seg0;          // [npeers] first send offset of the peer's segment
    double *const *dst;           // [npeers] peer's receive window at MY segment, slot 0
    const int64_t *dst_stride;    // [npeers] slot stride there (= the peer's ghost count)
    uint64_t *const *flag_dst;    // [npeers] my kHaloWG flags in the peer's window
    const uint64_t *ack;          // [npeers] local: last epoch peer p has finished reading (stored by p)
    // receiver side
    const int *nflag;             // [npeers] sender workgroups of peer p (0: nothing comes from p)
    const uint64_t *flags;        // local [npeers][kHaloWG]
    const double *rwin;           // local [2][n_ghost]
    uint64_t *const *ack_dst;     // [npeers] my word in peer p's ack array
    unsigned long long *arrive;   // local device words: [0] consumer workgroups done with the window, [1] workgroups of a
                                  // one-kernel exchange that have finished (both reset by the last arrival)
    uint64_t *epoch;              // local device word: exchanges completed.  Every workgroup of an exchange reads it when it
                                  // starts, so it may only move once ALL workgroups of the launch have started: a push
                                  // workgroup dispatched late - after this rank's consumers are done, which needs the
                                  // NEIGHBOURS' pushes only - would otherwise take the next epoch for its own and write
                                  // the other window slot (seen as stale ghosts when another process held the CUs)
    int *status;
    unsigned long long ticks;
    int64_t n_ghost;
    int npeers;
};


// Consumer side of an exchange whose ghost values are read straight from the receive window (no unpack):
//   halo_window_ready : ONE wave of the workgroup polls every neighbour's flags of epoch e (bounded), takes the system-scope
//                       acquire; the caller's barrier then releases the other waves;
//   halo_consumed     : every workgroup of the consuming launch calls it exactly once when it will read the window no more;
//                       the last one acknowledges to the senders and - when the launch holds consumers only - completes the
//                       epoch (a one-kernel exchange completes it through halo_launch_done instead).
__device__ __forceinline__ void halo_window_ready(const HaloPeerDev &H, uint64_t e) {
    if (threadIdx.x < 64) {
        SpinGuard guard(H.ticks, H.status, 3);
        for (int p = 0; p < H.npeers; ++p) {
            const bool need = (int)threadIdx.x < H.nflag[p];
            const uint64_t *f = H.flags + (size_t)p * kHaloWG + (threadIdx.x & (kHaloWG - 1));
            for (;;) {
                const bool ok = !need || ld_sys(f) >= e;
                if (__all(ok)) break;
                if (__any(guard.expired())) break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // system scope: nothing stale of the window in this CU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the invalidate has completed before the barrier opens
    }
}
__device__ __forceinline__ void halo_consumed(const HaloPeerDev &H, uint64_t e, unsigned nworkgroups, bool completes_epoch = true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = atomicAdd(H.arrive, 1ULL) + 1ULL;
        if (done == nworkgroups) {
            // last workgroup: the window slot is free again, tell the senders; the exchange is complete.  (Every other
            // workgroup has arrived and the next user of the counter is a later launch: a plain reset is safe.)
            *H.arrive = 0ULL;
            for (int p = 0; p < H.npeers; ++p)
                if (H.nflag[p] > 0) st_sys(H.ack_dst[p], e);
            if (completes_epoch) *H.epoch = e;
        }
    }
}
// one-kernel exchange: every workgroup of the launch calls it as its last action; the last one completes the epoch
__device__ __forceinline__ void halo_launch_done(const HaloPeerDev &H, uint64_t e, unsigned nworkgroups) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = atomicAdd(H.arrive + 1, 1ULL) + 1ULL;
        if (done == nworkgroups) {
            H.arrive[1] = 0ULL;
            *H.epoch = e;
        }
    }
}

}  // namespace npg
