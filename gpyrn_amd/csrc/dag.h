// Device side of the dataflow schedule (queue.hip): what a kernel does to wait for a node's inputs and to hand
// its outputs on.  Included by the worker kernel and by the latency chain's three kernels.
//
// Memory ordering (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   producer: every storing wave's s_waitcnt vmcnt(0), the workgroup's barrier, ONE lane's agent-scope release,
//             then that lane's atomics on the graph's counters and its stores to the ready queues;
//   consumer: one lane polls (relaxed), then ONE agent-scope acquire + s_waitcnt vmcnt(0), the workgroup's barrier,
//             plain loads.
// A counter that reaches zero is seen to do so by the workgroup whose decrement was last; that workgroup pushes the
// node.  The producers of the node's other inputs released before THEIR decrements, which precede this one in the
// counter's modification order, so their data is at the point of coherence before the entry can be read.
#pragma once
#include "gprn_internal.h"

#ifdef __HIPCC__
__device__ __forceinline__ unsigned q_load(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void q_store(unsigned* p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned* q_head(const QueueCtl& q, int c) { return q.ctr + c * GPRN_QCTR_STRIDE; }
__device__ __forceinline__ unsigned* q_tail(const QueueCtl& q, int c) { return q.ctr + (GPRN_QCLASSES + c) * GPRN_QCTR_STRIDE; }
__device__ __forceinline__ unsigned* q_left(const QueueCtl& q) { return q.ctr + QC_LEFT * GPRN_QCTR_STRIDE; }
__device__ __forceinline__ unsigned* q_bell(const QueueCtl& q) { return q.ctr + QC_BELL * GPRN_QCTR_STRIDE; }

// one lane: node `op` of matrix m has all its inputs -- onto the ready queue of its class, nent entries
__device__ __forceinline__ void q_push(const QueueCtl& q, unsigned m, unsigned op)
{
    const QOp* o = q.ops + op;
    const unsigned cls = o->cls, n = o->nent;
    const unsigned at = __hip_atomic_fetch_add(q_tail(q, cls), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == 1) q_store(q.slots[cls] + at, q_entry(m, GPRN_Q_WHOLE, op));
    else {
        const bool syrk = o->flags & QF_DIAG_SYRK;         // quarters 0, 2, 3 only
        for (unsigned e = 0; e < n; ++e) q_store(q.slots[cls] + at + e, q_entry(m, syrk && e ? e + 1 : e, op));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the entries have landed before the bell rings
    // the doorbell: ONE word that idle workers watch, instead of every class's head and tail (hundreds of idle
    // workgroups polling ten lines each every few microseconds slowed every kernel on the chip two- to threefold)
    __hip_atomic_fetch_add(q_bell(q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a wait of the schedule: *p & mask == 0 within the budget; false when it (or an earlier one) gave up
__device__ __forceinline__ bool q_spin_zero(const unsigned* p, unsigned mask, unsigned* timed_out, unsigned at_most = 0)
{
    if ((q_load(p) & mask) <= at_most) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = timed_out[1];
    for (;;) {
        __builtin_amdgcn_s_sleep(32);                      // ~1 us: up to 36 workgroups per matrix poll the same word
        if ((q_load(p) & mask) <= at_most) return true;
        if (q_load(timed_out)) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget) { atomicExch(timed_out, 1u); return false; }
    }
}

// GPRN_QUEUE_TRACE: one record of the call's timeline -- who (a queue entry and the worker, or a chain node), and three
// stamps of the 100 MHz clock.  One lane calls it.
__device__ __forceinline__ void q_trace(const QueueCtl& q, unsigned long long who, unsigned long long t1, unsigned long long t2,
                                        unsigned long long t3)
{
    if (!q.trace) return;
    const unsigned long long at = atomicAdd(q.trace, 1ull);
    if (at >= (unsigned long long)q.trace_cap) return;
    unsigned long long* r = q.trace + 1 + 4 * at;
    r[0] = who | ((unsigned long long)(unsigned)q.call_id << 56);
    r[1] = t1; r[2] = t2; r[3] = t3;
}

// start of a chain kernel: every thread of the workgroup calls it; returns once the node's inputs are there
__device__ __forceinline__ void q_await(const QueueCtl& q, unsigned m, unsigned op)
{
    if (threadIdx.x == 0) {
        (void)q_spin_zero(q.state + (size_t)m * q.nops + op, 0xffffu, q.timed_out);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// end of one entry of node `op` of matrix m (a workgroup of a chain kernel, a worker's sub-tile or whole node):
// every thread of the workgroup calls it.  The entry that finishes the node tells the successors.
__device__ __forceinline__ void q_complete(const QueueCtl& q, unsigned m, unsigned op, bool counted_in_left)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    unsigned* const st = q.state + (size_t)m * q.nops;
    unsigned last = 0;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_sub(st + op, 1u << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old >> 16) == 1u;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) {
        const unsigned s0 = q.ops[op].succ0, ns = q.ops[op].nsucc;
        for (unsigned base = 0; base < ns; base += 64) {
            const unsigned i = base + lane;
            if (i < ns) {
                const unsigned s = q.succ[s0 + i];
                const unsigned old = __hip_atomic_fetch_sub(st + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((old & 0xffffu) == 1u && q.ops[s].kind != QK_CHAIN) q_push(q, m, s);
            }
        }
    }
    // (after the pushes: a worker that sees `left` reach zero may leave, and nothing is pushed after the last entry)
    if (lane == 0 && counted_in_left)
        __hip_atomic_fetch_sub(q_left(q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
