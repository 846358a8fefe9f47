// Dataflow schedule of the blocked factorisation  B = L L^T, X = L^-1  (factor.hip has the algebra).
//
// Replaces, for the same reference code as factor.hip (meanfield.py:771,850,1087-1090,1041,1051 and
// _cholNugget :71-89), the launch-per-step schedule of factor_invert_split: the factorisation of a batch is ONE
// task graph per matrix,
//     chain nodes   diag(k), L_{k+1,k}, B_{k+1,k+1} update      -- the three latency-critical kernels of a tile step,
//                                                                   still launches on the chain stream (factor.hip,
//                                                                   gemm_tile.hip), each polling its own node;
//     tile nodes    every other 128 x 128 product of the algorithm (panel products, in-panel K = 128 updates,
//                   K = 512 outer updates, and -- as filler -- the previous phase's X^T X product),
// executed by ONE persistent kernel of 2 workgroups per CU (k_tile_queue) that pulls ready nodes from priority
// queues.  What the launch schedule loses and this one does not:
//   * false dependencies: a launch waits for whole launches (the chain's L_{k+2,k+1} for ALL of step k's in-panel
//     updates, hence for the previous panel's "next" outer update: 70-90 us at every panel boundary of a 2-matrix
//     phase, profiles/r02_chain_timeline_cfg3.txt); a node waits for the tiles it reads;
//   * priorities: hardware dispatch is first come first served and workgroups are never preempted, so a panel
//     launch queues behind 26-40 us bulk workgroups; a worker that finishes any task takes the most urgent ready
//     one next;
//   * launch tails and the per-step chain of dependent launches + synchronisation kernels on the side stream.
// The graph is derived on the host from the SEQUENTIAL algorithm (the same task formulas as ensure_tasks): walking
// the operations in program order, every operation gets an edge from the last writer of each tile it reads
// (read-after-write), from the last writer of the tile it writes (write order) and from every reader of that tile
// since (write-after-read).  Any execution that respects the edges computes what the sequential program computes.
#include "gprn_internal.h"
#include "dag.h"
#include "tile_mma.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>

struct QueuePlanRef {
    int T = 0, outer = 0;
    std::vector<QOp> ops;                  // main graph, then the X^T X nodes (no edges)
    std::vector<uint32_t> succ, init;      // CSR successors; initial state word per node
    std::vector<uint32_t> diag_op, l_op, u_op;   // chain nodes per tile step (~0u: none)
    uint32_t n_main = 0, lauum0 = 0, n_lauum = 0;
    uint32_t ent_main[GPRN_QCLASSES] = {0}, ent_lauum = 0;     // ready-queue entries per matrix and class
    double flops_main = 0.0, flops_lauum = 0.0;                // per matrix (tile nodes; diagonal SYRK tiles count 3/4)
    QOp* d_ops = nullptr;
    uint32_t *d_succ = nullptr, *d_init = nullptr;
};

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ------------------------------------------------------------------ graph construction (host)
namespace {

struct TileRef { int buf, i, j; };

static inline int64_t toffq(int ti, int tj, int ld) { return ((int64_t)ti * GPRN_TILE) * ld + (int64_t)tj * GPRN_TILE; }

struct Builder {
    int T, ld;
    std::vector<QOp>& ops;
    std::vector<std::vector<uint32_t>> preds;      // per op (deduplicated)
    std::vector<int32_t> last_writer;              // per tile
    std::vector<std::vector<uint32_t>> readers;    // per tile, since the last write
    std::vector<uint32_t> stamp;
    Builder(int T_, int ld_, std::vector<QOp>& o) : T(T_), ld(ld_), ops(o), last_writer(2 * T_ * T_, -1), readers(2 * T_ * T_) {}
    int tid(int buf, int i, int j) const { return (buf == BUF_B ? 0 : 1) * T * T + i * T + j; }
    void edge(uint32_t from, uint32_t to) {
        if (from == to) return;
        auto& p = preds[to];
        if (std::find(p.begin(), p.end(), from) == p.end()) p.push_back(from);
    }
    // the operation reads `rd` and writes `wr` (tiles); returns its index
    uint32_t add(const QOp& o, const std::vector<TileRef>& rd, const std::vector<TileRef>& wr) {
        const uint32_t id = (uint32_t)ops.size();
        ops.push_back(o);
        preds.emplace_back();
        for (const TileRef& r : rd) {
            const int t = tid(r.buf, r.i, r.j);
            if (last_writer[t] >= 0) edge((uint32_t)last_writer[t], id);
        }
        for (const TileRef& w : wr) {
            const int t = tid(w.buf, w.i, w.j);
            if (last_writer[t] >= 0) edge((uint32_t)last_writer[t], id);
            for (uint32_t r : readers[t]) edge(r, id);
        }
        for (const TileRef& r : rd) readers[tid(r.buf, r.i, r.j)].push_back(id);
        for (const TileRef& w : wr) {
            const int t = tid(w.buf, w.i, w.j);
            last_writer[t] = (int32_t)id;
            readers[t].clear();
        }
        return id;
    }
    // tiles a TileTask reads and the one it writes
    void tiles_of(const TileTask& t, std::vector<TileRef>& rd, TileRef& wr) const {
        rd.clear();
        const int64_t row = (int64_t)GPRN_TILE * ld;
        const int nk = t.klen / GPRN_TILE;
        const int a_mode = (t.modes >> 2) & 1, b_mode = (t.modes >> 3) & 1, c_mode = t.modes & 3;
        const int ai = (int)(t.a_off / row), aj = (int)((t.a_off % ld) / GPRN_TILE);
        const int bi = (int)(t.b_off / row), bj = (int)((t.b_off % ld) / GPRN_TILE);
        for (int k = 0; k < nk; ++k) {
            rd.push_back(a_mode == 0 ? TileRef{t.a_buf, ai, aj + k} : TileRef{t.a_buf, ai + k, aj});
            rd.push_back(b_mode == 0 ? TileRef{t.b_buf, bi, bj + k} : TileRef{t.b_buf, bi + k, bj});
        }
        wr = TileRef{t.c_buf, (int)(t.c_off / row), (int)((t.c_off % ld) / GPRN_TILE)};
        if (c_mode == CM_SUB) rd.push_back(wr);
    }
};

}   // namespace

// priority class of a tile node created at tile step k_cur that writes tile (i, j) of buffer buf
static int class_of(int buf, int i, int j, int k_cur, bool chain_tile)
{
    const int deadline = buf == BUF_B ? j : i;         // the tile step that consumes the tile (column / row of the inverse)
    const int slack = deadline - k_cur;
    if (chain_tile && slack <= 2) return 0;
    if (slack <= 1) return 1;
    if (slack <= 4) return 2;
    return 3;
}

static QueuePlanRef* build_plan_host(int T, int ld, int outer)
{
    QueuePlanRef* P = new QueuePlanRef();
    P->T = T; P->outer = outer;
    Builder b(T, ld, P->ops);
    P->diag_op.assign(T, ~0u); P->l_op.assign(T, ~0u); P->u_op.assign(T, ~0u);
    const int whole_from = env_int("GPRN_QUEUE_WHOLE_FROM", 3);    // classes from this one on: one entry per node
    const int syrk_skip = env_int("GPRN_QUEUE_SYRK", 1);
    std::vector<TileRef> rd;
    TileRef wr;
    auto tile_node = [&](const TileTask& t, int kind, int k_cur, bool chain_tile) {
        b.tiles_of(t, rd, wr);
        QOp o;
        memset(&o, 0, sizeof(o));
        o.t = t;
        o.kind = (uint8_t)kind;
        o.cls = (uint8_t)class_of(wr.buf, wr.i, wr.j, k_cur, chain_tile);
        const bool syrk = syrk_skip && kind == QK_TILE && t.c_buf == BUF_B && wr.i == wr.j && t.a_buf == t.b_buf &&
                          t.a_off == t.b_off && ((t.modes >> 2) & 3) == 0;
        o.flags = syrk ? QF_DIAG_SYRK : 0;
        const int nsub = kind == QK_TILE ? (syrk ? 3 : 4) : 2;
        o.nent = (uint8_t)(o.cls >= whole_from ? 1 : nsub);
        P->flops_main += 2.0 * GPRN_TILE * GPRN_TILE * (double)t.klen * (kind == QK_TILE ? (syrk ? 0.75 : 1.0) : 0.5625);
        return b.add(o, rd, {wr});
    };
    auto chain_node = [&](int type, int k, int nent, const std::vector<TileRef>& r, const std::vector<TileRef>& w) {
        QOp o;
        memset(&o, 0, sizeof(o));
        o.kind = QK_CHAIN;
        o.nent = (uint8_t)nent;
        o.t.c_buf = (uint8_t)type;               // 0 diag(k), 1 L_{k+1,k}, 2 update of B_{k+1,k+1}: for the host-side view only
        o.t.klen = k;
        return b.add(o, r, w);
    };
    for (int k0 = 0; k0 < T; k0 += outer) {
        const int k1 = std::min(T, k0 + outer);
        for (int k = k0; k < k1; ++k) {
            // ---- the chain's three kernels of the step
            P->diag_op[k] = chain_node(0, k, 1, {TileRef{BUF_B, k, k}}, {TileRef{BUF_B, k, k}, TileRef{BUF_X, k, k}});
            if (k + 1 < T)
                P->l_op[k] = chain_node(1, k, GPRN_TILE / 16, {TileRef{BUF_X, k, k}, TileRef{BUF_B, k + 1, k}}, {TileRef{BUF_B, k + 1, k}});
            // ---- panel: L_ik = B_ik X_kk^T (i > k + 1; in place), X_kc = X_kk R_kc (c < k; in place)
            for (int i = k + 2; i < T; ++i)
                tile_node(TileTask{toffq(i, k, ld), toffq(i, k, ld), toffq(k, k, ld), GPRN_TILE, BUF_B, BUF_B, BUF_X,
                                   tile_modes(CM_SET, 0, 0)}, QK_PANEL_L, k, i == k + 2);
            for (int cc = 0; cc < k; ++cc)
                tile_node(TileTask{toffq(k, cc, ld), toffq(k, k, ld), toffq(k, cc, ld), GPRN_TILE, BUF_X, BUF_X, BUF_X,
                                   tile_modes(CM_SET, 0, 1)}, QK_PANEL_X, k, false);
            if (k + 1 < T)
                P->u_op[k] = chain_node(2, k, 36, {TileRef{BUF_B, k + 1, k}, TileRef{BUF_B, k + 1, k + 1}}, {TileRef{BUF_B, k + 1, k + 1}});
            // ---- in-panel updates, K = 128: the panel's own columns right of k, and of the NEXT panel the diagonal
            // and sub-diagonal tiles (what the chain works on there; ensure_tasks, factor.hip)
            for (int j = k + 1; j < std::min(T, k1 + outer); ++j)
                for (int i = j; i < (j < k1 ? T : std::min(T, j + 2)); ++i) {
                    if (i == k + 1 && j == k + 1) continue;               // the chain's own update
                    tile_node(TileTask{toffq(i, j, ld), toffq(i, k, ld), toffq(j, k, ld), GPRN_TILE, BUF_B, BUF_B, BUF_B,
                                       tile_modes(CM_SUB, 0, 0)}, QK_TILE, k, i <= j + 2);
                }
            for (int i = k + 1; i < k1; ++i)
                for (int cc = 0; cc <= k; ++cc)
                    tile_node(TileTask{toffq(i, cc, ld), toffq(i, k, ld), toffq(k, cc, ld), GPRN_TILE, BUF_X, BUF_B, BUF_X,
                                       tile_modes(cc == k ? CM_SETNEG : CM_SUB, 0, 1)}, QK_TILE, k, false);
        }
        // ---- outer update of the panel, K = its width
        const int kw = (k1 - k0) * GPRN_TILE;
        const int n1 = std::min(T, k1 + outer);
        for (int i = k1; i < T; ++i) {
            for (int j = k1; j <= i; ++j) {
                if (j < n1 && i <= j + 1) continue;                       // kept up to date step by step
                tile_node(TileTask{toffq(i, j, ld), toffq(i, k0, ld), toffq(j, k0, ld), kw, BUF_B, BUF_B, BUF_B,
                                   tile_modes(CM_SUB, 0, 0)}, QK_TILE, k1 - 1, i <= j + 2);
            }
            for (int cc = 0; cc < k0; ++cc)
                tile_node(TileTask{toffq(i, cc, ld), toffq(i, k0, ld), toffq(k0, cc, ld), kw, BUF_X, BUF_B, BUF_X,
                                   tile_modes(CM_SUB, 0, 1)}, QK_TILE, k1 - 1, false);
            for (int cc = k0; cc < k1; ++cc)
                tile_node(TileTask{toffq(i, cc, ld), toffq(i, cc, ld), toffq(cc, cc, ld), (k1 - cc) * GPRN_TILE, BUF_X, BUF_B,
                                   BUF_X, tile_modes(CM_SETNEG, 0, 1)}, QK_TILE, k1 - 1, false);
        }
    }
    P->n_main = (uint32_t)P->ops.size();
    // ---- successors (CSR) and initial state words of the main graph
    std::vector<uint32_t> nsucc(P->n_main, 0);
    for (uint32_t v = 0; v < P->n_main; ++v)
        for (uint32_t u : b.preds[v]) nsucc[u] += 1;
    uint32_t at = 0;
    for (uint32_t v = 0; v < P->n_main; ++v) { P->ops[v].succ0 = at; P->ops[v].nsucc = 0; at += nsucc[v]; }
    P->succ.assign(at, 0);
    for (uint32_t v = 0; v < P->n_main; ++v)
        for (uint32_t u : b.preds[v]) P->succ[P->ops[u].succ0 + P->ops[u].nsucc++] = v;
    P->init.resize(P->n_main);
    for (uint32_t v = 0; v < P->n_main; ++v) {
        P->init[v] = ((uint32_t)P->ops[v].nent << 16) | (uint32_t)b.preds[v].size();
        if (P->ops[v].kind != QK_CHAIN) P->ent_main[P->ops[v].cls] += P->ops[v].nent;
    }
    // ---- X^T X of the previous phase (BUF_B = lower(X^T X)): no edges, last class, whole nodes
    P->lauum0 = (uint32_t)P->ops.size();
    for (int a = T - 1; a >= 0; --a)
        for (int bb = 0; bb <= a; ++bb) {
            QOp o;
            memset(&o, 0, sizeof(o));
            o.t = TileTask{toffq(a, bb, ld), toffq(a, a, ld), toffq(a, bb, ld), ld - a * GPRN_TILE, BUF_B, BUF_X, BUF_X,
                           tile_modes(CM_SET, 1, 1)};
            o.kind = QK_TILE;
            o.cls = GPRN_QCLASSES - 1;
            o.nent = 1;
            P->ops.push_back(o);
            P->init.push_back(1u << 16);
            P->flops_lauum += 2.0 * GPRN_TILE * GPRN_TILE * (double)o.t.klen;
        }
    P->n_lauum = (uint32_t)P->ops.size() - P->lauum0;
    P->ent_lauum = P->n_lauum;
    return P;
}

static QueuePlanRef* build_plan(gprn_ctx* c, int T, int ld, int outer)
{
    QueuePlanRef* P = build_plan_host(T, ld, outer);
    auto up = [&](auto** d, const auto& v) {
        using E = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void**)d, std::max<size_t>(1, v.size()) * sizeof(E)) != hipSuccess) return false;
        return v.empty() || hipMemcpy(*d, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&P->d_ops, P->ops) || !up(&P->d_succ, P->succ) || !up(&P->d_init, P->init)) {
        c->err = "queue plan: device allocation failed";
        if (P->d_ops) hipFree(P->d_ops);
        if (P->d_succ) hipFree(P->d_succ);
        if (P->d_init) hipFree(P->d_init);
        delete P;
        return nullptr;
    }
    return P;
}

static void plan_free(QueuePlanRef* P)
{
    if (!P) return;
    if (P->d_ops) hipFree(P->d_ops);
    if (P->d_succ) hipFree(P->d_succ);
    if (P->d_init) hipFree(P->d_init);
    delete P;
}

// GPRN_QUEUE_STATS=1: what the workers did since the last call of this, summed over the workers (printed by
// gprn_destroy and by gprn_profile_read)
// GPRN_QUEUE_TRACE=n: the records so far go to $GPRN_QUEUE_TRACE_FILE (default gpurun_out/queue_trace.bin): a header
// {magic, records, nodes of the last plan}, the records (4 x u64), then per node of that plan 4 x i32 (kind, class, K,
// C tile as buf << 16 | i << 8 | j) -- profiles/queue_timeline.py reads it
static void queue_dump_trace(gprn_ctx* c)
{
    if (!c->d_qtrace) return;
    (void)hipDeviceSynchronize();
    unsigned long long n = 0;
    if (hipMemcpy(&n, c->d_qtrace, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess || n == 0) return;
    n = std::min<unsigned long long>(n, (unsigned long long)c->qtrace_cap);
    std::vector<unsigned long long> rec((size_t)n * 4);
    if (hipMemcpy(rec.data(), c->d_qtrace + 1, rec.size() * sizeof(rec[0]), hipMemcpyDeviceToHost) != hipSuccess) return;
    (void)hipMemset(c->d_qtrace, 0, sizeof(unsigned long long));
    const QueuePlanRef* P = c->qplan[0] ? c->qplan[0] : c->qplan[1];
    const char* path = getenv("GPRN_QUEUE_TRACE_FILE");
    FILE* f = fopen(path ? path : "gpurun_out/queue_trace.bin", "wb");
    if (!f) return;
    const unsigned long long head[4] = {0x47505251ull, n, P ? (unsigned long long)P->ops.size() : 0ull, P ? (unsigned long long)P->T : 0ull};
    fwrite(head, sizeof(head), 1, f);
    fwrite(rec.data(), sizeof(rec[0]), rec.size(), f);
    if (P)
        for (const QOp& o : P->ops) {
            const int64_t row = (int64_t)GPRN_TILE * P->T * GPRN_TILE;
            const int ld = P->T * GPRN_TILE;
            int32_t v[4] = {o.kind, o.cls, o.t.klen,
                            o.kind == QK_CHAIN ? (int32_t)o.t.c_buf : (int32_t)((o.t.c_buf << 16) | ((o.t.c_off / row) << 8) | ((o.t.c_off % ld) / GPRN_TILE))};
            fwrite(v, sizeof(v), 1, f);
        }
    fclose(f);
    fprintf(stderr, "[gprn] queue trace: %llu records written\n", n);
}

void queue_print_stats(gprn_ctx* c)
{
    queue_dump_trace(c);
    if (!c->d_qstats) return;
    std::vector<unsigned long long> h((size_t)4096 * 16);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(h.data(), c->d_qstats, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost) != hipSuccess) return;
    (void)hipMemset(c->d_qstats, 0, h.size() * sizeof(h[0]));
    double tot[16] = {0};
    int nw = 0;
    for (int w = 0; w < 4096; ++w) {
        if (!h[(size_t)w * 16 + 6]) continue;
        ++nw;
        for (int i = 0; i < 16; ++i) tot[i] += (double)h[(size_t)w * 16 + i];
    }
    if (!nw || tot[4] == 0) return;
    const double us = 0.01;                        // 100 MHz ticks
    const double all = tot[0] + tot[1] + tot[2] + tot[3] + tot[7] + tot[8];
    fprintf(stderr, "[gprn] queue workers: %d workgroups x %.0f launches, %.0f entries; per entry us: idle %.2f  claim %.2f  acquire %.2f  "
                    "node+pointers %.2f  contraction %.2f  completion %.2f  (K per entry %.0f); share of worker time: idle %.1f %%, "
                    "contraction %.1f %%\n",
            nw, tot[6] / nw, tot[4], tot[0] * us / tot[4], tot[7] * us / tot[4], tot[1] * us / tot[4], tot[8] * us / tot[4],
            tot[2] * us / tot[4], tot[3] * us / tot[4], 16.0 * tot[5] / tot[4], 100.0 * tot[0] / all, 100.0 * tot[2] / all);
}

void queue_free(gprn_ctx* c)
{
    queue_print_stats(c);
    if (c->d_qstats) { hipFree(c->d_qstats); c->d_qstats = nullptr; }
    if (c->d_qtrace) { hipFree(c->d_qtrace); c->d_qtrace = nullptr; }
    for (int s = 0; s < 2; ++s) { plan_free(c->qplan[s]); c->qplan[s] = nullptr; }
    if (c->d_qstate) hipFree(c->d_qstate);
    if (c->d_qslots) hipFree(c->d_qslots);
    if (c->d_qctr) hipFree(c->d_qctr);
    c->d_qstate = nullptr; c->d_qslots = nullptr; c->d_qctr = nullptr;
    c->qstate_cap = c->qslots_cap = 0;
    if (c->ev_qreset) hipEventDestroy(c->ev_qreset);
    if (c->ev_qdone) hipEventDestroy(c->ev_qdone);
    c->ev_qreset = c->ev_qdone = nullptr;
}

// ------------------------------------------------------------------ device side
// State of a call: state words from the plan's initial ones, every slot empty, counters zero, `left` = the number of
// ready-queue entries the workers will consume; the X^T X nodes (no inputs) go straight into the last queue.
__global__ void k_queue_init(QueueCtl q, QueueCtl* __restrict__ qimg, const uint32_t* __restrict__ init, int n_main, int nbatch,
                             int lauum0, int n_lauum, int n_extra, unsigned left_total, int hold_op)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *qimg = q;        // the workers read the control block from memory
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t per = (size_t)q.nops;
    for (size_t i = tid; i < (size_t)(nbatch + n_extra) * per; i += nth) {
        const size_t m = i / per, op = i % per;
        const bool live = m < (size_t)nbatch ? op < (size_t)n_main : op >= (size_t)lauum0;
        // (hold_op >= 0, test hook: that node of matrix 0 gets a dependency nobody will ever meet)
        q.state[i] = live ? init[op] + (m == 0 && (int)op == hold_op ? 1u : 0u) : 0u;
    }
    for (int cl = 0; cl < GPRN_QCLASSES; ++cl)
        for (size_t i = tid; i < q.cap[cl]; i += nth) {
            unsigned v = GPRN_Q_EMPTY;
            if (cl == GPRN_QCLASSES - 1 && i < (size_t)n_extra * n_lauum)
                v = q_entry((unsigned)(nbatch + i / n_lauum), GPRN_Q_WHOLE, (unsigned)(lauum0 + i % n_lauum));
            q.slots[cl][i] = v;
        }
    if (tid < QC_COUNT && tid != QC_TIMEOUT) {
        unsigned v = 0;
        if (tid == QC_LEFT) v = left_total;
        if (tid == QC_TAIL + GPRN_QCLASSES - 1) v = (unsigned)n_extra * (unsigned)n_lauum;       // tail of the last class
        if (tid == QC_XCC) for (int x = 1; x < 8; ++x) q.ctr[tid * GPRN_QCTR_STRIDE + x] = 0u;
        q.ctr[tid * GPRN_QCTR_STRIDE] = v;
    }
    for (size_t i = tid; i < GPRN_QCU_WORDS; i += nth) q.ctr[QC_COUNT * GPRN_QCTR_STRIDE + i] = 0u;
}

// One-wave wait on a stream, in front of a chain launch whose workgroups would otherwise all poll: lane m waits for
// node `op` of matrix m.
// (op2: a second node that may start once all but `left2` of its inputs are there -- the B_{k+1,k+1} update, whose
// last input is the L_{k+1,k} launch right in front of it on the stream)
__global__ void k_queue_wait(QueueCtl q, unsigned op, int nbatch, unsigned op2, unsigned left2)
{
    for (int m = threadIdx.x; m < nbatch; m += 64) {
        (void)q_spin_zero(q.state + (size_t)m * q.nops + op, 0xffffu, q.timed_out);
        if (op2 != ~0u) (void)q_spin_zero(q.state + (size_t)m * q.nops + op2, 0xffffu, q.timed_out, left2);
    }
}

// The next ready entry for this workgroup (its first wave calls it, all 64 lanes), or GPRN_Q_EMPTY when every entry
// of the call is done (or a wait of the call gave up).  Classes in priority order.
//   * A slot goes EMPTY -> entry (the pusher's store) -> TAKEN (the taker's exchange), and the exchange IS the claim:
//     nobody is ever promised a slot that is not filled yet.  (A first version handed out tickets with a fetch-add on
//     the head: a burst of two entries seen by a hundred idle workers left ninety-eight of them holding tickets for the
//     NEXT entries of that class -- which then waited for their particular holder to finish whatever long task it had
//     picked up meanwhile: a priority inversion built into the queue.)
//   * The wave looks at 64 consecutive slots at once (one coalesced load); workers start at different lanes, and with a
//     long queue at different 64-slot windows, so that a burst is handed out in parallel rather than one
//     compare-and-swap at a time.
//   * head is a hint: every slot below it is TAKEN.  A worker that finds TAKEN slots at the head moves it (atomic max).
__device__ __forceinline__ unsigned q_claim(const QueueCtl& q, unsigned wid, unsigned long long* t_pass = nullptr)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = q.timed_out[1];
    const unsigned rot = (wid * 2654435761u) >> 26;                 // this worker's first lane, 0..63
    unsigned seen_bell = q_load(q_bell(q));
    for (unsigned pass = 0;; ++pass) {
        if (t_pass) *t_pass = __builtin_amdgcn_s_memrealtime();
        unsigned hv = 0;
        if (lane < GPRN_QCLASSES) hv = q_load(q_head(q, lane));
        else if (lane >= 8 && lane < 8 + GPRN_QCLASSES) hv = q_load(q_tail(q, lane - 8));
        for (int c = 0; c < GPRN_QCLASSES; ++c) {
            const unsigned h = __builtin_amdgcn_readlane(hv, c), tl = __builtin_amdgcn_readlane(hv, 8 + c);
            if (h >= tl) continue;
            const unsigned nwin = min(4u, (tl - h + 63u) >> 6);
            const unsigned w = nwin > 1 ? (wid + pass) % nwin : 0u;
            const unsigned base = h + 64u * w;
            const unsigned v = base + lane < tl ? q_load(q.slots[c] + base + lane) : GPRN_Q_EMPTY;
            unsigned long long mask = __ballot(v < GPRN_Q_TAKEN);
            if (w == 0) {
                const unsigned long long taken = __ballot(v == GPRN_Q_TAKEN);
                const unsigned lead = ~taken ? (unsigned)__builtin_ctzll(~taken) : 64u;
                if (lead && lane == 0) atomicMax(q_head(q, c), h + lead);
            }
            while (mask) {
                const unsigned long long r = rot ? (mask >> rot) | (mask << (64 - rot)) : mask;
                const unsigned b = ((unsigned)__builtin_ctzll(r) + rot) & 63u;
                unsigned got = GPRN_Q_TAKEN;
                if (lane == (int)b)
                    got = __hip_atomic_exchange(q.slots[c] + base + b, GPRN_Q_TAKEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = __builtin_amdgcn_readlane(got, b);
                if (got < GPRN_Q_TAKEN) return got;
                mask &= ~(1ull << b);
            }
        }
        if (q_load(q_left(q)) == 0u) return GPRN_Q_EMPTY;
        if (q_load(q.timed_out)) return GPRN_Q_EMPTY;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
            // nothing became ready for a whole budget: the producers are gone (a serialising tool, a lost launch)
            if (lane == 0) atomicExch(q.timed_out, 1u);
            return GPRN_Q_EMPTY;
        }
        // Nothing for this workgroup: nap until the doorbell moves (or a while has passed -- entries that were being
        // claimed by others when this pass looked may have been left).  Naps are long: what matters is the polling rate of
        // ALL idle workgroups together (hundreds of them each reading a line every few microseconds slowed every kernel on
        // the chip two- to threefold, MI355X_MICROARCH.md "polling-cost"), and with many asleep at different phases a new
        // entry is still seen within a fraction of one nap.
        if (pass < 2) { __builtin_amdgcn_s_sleep(16); continue; }
        const unsigned bell = q_load(q_bell(q));
        if (bell != seen_bell) { seen_bell = bell; continue; }
        for (int nap = 0; nap < 8; ++nap) {
            __builtin_amdgcn_s_sleep(127);                       // ~3.4 us
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
            const unsigned b2 = q_load(q_bell(q));
            if (b2 != seen_bell) { seen_bell = b2; break; }
        }
    }
}

// One 64 x 64 part of a tile node.  QK_TILE: quarter `sub` (row half sub / 2, column half sub % 2).  The panel
// products are in place, so a workgroup owns 64 whole rows (L_ik = B_ik X_kk^T: the tile is its own A operand)
// resp. 64 whole columns (X_kc = X_kk R_kc: its own B operand) and computes them in two passes, the half that needs
// all of K first: with the triangular X_kk that half only overwrites what the second pass does not read (k < 64).
// One node (or one part of it) as a sequence of 64 x 64 contractions.  Every triangular form of tile_mma appears at
// ONE place in the kernel (inlined at each call site the kernel needed 150-170 VGPRs -- the budget is 96, see
// k_tile_queue -- for code that is never live at the same time), inside a loop over "pieces" that the parts of all
// three node kinds are reduced to.
struct Piece { const double* A; const double* B; gptr_t C; int klen, mb, nb; };

__device__ __forceinline__ Piece piece_of(const TileTask& t, int kind, int sub, int pass, double* const (&gp)[GPRN_NBUF], int ld)
{
    auto pick = [&](int b) { return b == 0 ? gp[0] : (b == 1 ? gp[1] : (b == 2 ? gp[2] : gp[3])); };
    const int a_mode = (t.modes >> 2) & 1, b_mode = (t.modes >> 3) & 1;
    const double* A0 = pick(t.a_buf) + t.a_off;
    const double* B0 = pick(t.b_buf) + t.b_off;
    gptr_t C0 = (gptr_t)(pick(t.c_buf) + t.c_off);
    Piece p;
    if (kind == QK_TILE) {
        const int sr = sub >> 1, sc = sub & 1;
        p.A = A0 + (a_mode ? (size_t)sr * 64 : (size_t)sr * 64 * ld);
        p.B = B0 + (b_mode ? (size_t)sc * 64 : (size_t)sc * 64 * ld);
        p.C = C0 + (size_t)sr * 64 * ld + sc * 64;
        p.klen = t.klen; p.mb = sr * 4; p.nb = sc * 4;
    } else if (kind == QK_PANEL_L) {
        // rows [64 sub, 64 sub + 64) of L_ik = B_ik X_kk^T (in place: the tile is its own A operand): pass 0 the columns
        // 64..127 over k < 128, pass 1 the columns 0..63 over k < 64 -- X_kk is lower triangular, so the first pass only
        // overwrites what the second does not read
        const int hi = pass == 0;
        p.A = A0 + (size_t)sub * 64 * ld;
        p.B = B0 + (hi ? (size_t)64 * ld : 0);
        p.C = C0 + (size_t)sub * 64 * ld + (hi ? 64 : 0);
        p.klen = hi ? GPRN_TILE : 64; p.mb = sub * 4; p.nb = hi ? 4 : 0;
    } else {
        // columns [64 sub, 64 sub + 64) of X_kc = X_kk R_kc (in place: its own B operand): rows 64..127 over k < 128, then
        // rows 0..63 over k < 64
        const int hi = pass == 0;
        p.A = A0 + (hi ? (size_t)64 * ld : 0);
        p.B = B0 + (size_t)sub * 64;
        p.C = C0 + (hi ? (size_t)64 * ld : 0) + sub * 64;
        p.klen = hi ? GPRN_TILE : 64; p.mb = hi ? 4 : 0; p.nb = sub * 4;
    }
    return p;
}

// All parts of one queue entry.  (Tried out of line, to keep the contraction's registers apart from the loop's: the
// arguments then travel in VGPRs, and even declared uniform again with readfirstlane the call cost ~25 us per entry.)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int64_t uni(int64_t v)
{
    const unsigned lo = uni((unsigned)(uint64_t)v), hi = uni((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <typename P>
__device__ __forceinline__ P* uni(P* p) { return (P*)(uintptr_t)uni((int64_t)(uintptr_t)p); }

__device__ __forceinline__ void run_node(double* lds_, const TileTask t_, int kind_, int flags_, unsigned sub_,
                                                   double* g0, double* g1, double* g2, double* g3, int ld_)
{
    double* const lds = uni(lds_);
    TileTask t;
    t.c_off = uni(t_.c_off); t.a_off = uni(t_.a_off); t.b_off = uni(t_.b_off); t.klen = uni(t_.klen);
    {
        const unsigned packed = uni((unsigned)t_.c_buf | ((unsigned)t_.a_buf << 8) | ((unsigned)t_.b_buf << 16) | ((unsigned)t_.modes << 24));
        t.c_buf = (uint8_t)packed; t.a_buf = (uint8_t)(packed >> 8); t.b_buf = (uint8_t)(packed >> 16); t.modes = (uint8_t)(packed >> 24);
    }
    const int kind = uni(kind_), flags = uni(flags_), ld = uni(ld_);
    const unsigned sub = uni(sub_);
    double* const gp[GPRN_NBUF] = {uni(g0), uni(g1), uni(g2), uni(g3)};
    const int c_mode = t.modes & 3, a_mode = (t.modes >> 2) & 1, b_mode = (t.modes >> 3) & 1;
    const int first = sub != GPRN_Q_WHOLE ? (int)sub : 0, end = sub != GPRN_Q_WHOLE ? (int)sub + 1 : (kind == QK_TILE ? 4 : 2);
    const int npass = kind == QK_TILE ? 1 : 2;
    bool again = false;
#pragma unroll 1
    for (int s2 = first; s2 < end; ++s2) {
        if (sub == GPRN_Q_WHOLE && (flags & QF_DIAG_SYRK) && s2 == 1) continue;
#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
            if (again) __syncthreads();                // the LDS stages are reused
            again = true;
            const Piece p = piece_of(t, kind, s2, pass, gp, ld);
            // (one instantiation for all three kinds: the panel products multiply X_kk's explicit zeros above the
            // diagonal instead of skipping them -- +1/3 on 5 % of the flops -- because the two triangular forms
            // cost 20 VGPRs more)
            tile_mma<64, 64, 2, 2, 0, false>(lds, p.A, p.B, p.C, ld, a_mode, b_mode, c_mode, p.klen, p.mb, p.nb);
        }
    }
}

// rows of the pointer table for the matrices behind the batch (the previous phase's X^T X), as a kernel argument
#define GPRN_Q_EXTRA 16
struct QExtra { double* p[GPRN_Q_EXTRA][GPRN_NBUF]; };

// (four waves per SIMD as the register budget: 128 VGPRs without a spill -- at 96, which would let two of these
// workgroups share a CU with a diagonal-block workgroup, the contraction spills; the launch leaves a few CUs with ONE
// worker instead, see factor_invert_queue)
// stats (GPRN_QUEUE_STATS=1, else null): per workgroup 8 words -- 100 MHz ticks spent looking for an entry, in the
// acquire, in the contractions, in the completion; entries done; K summed over them / 16
template <bool STATS>
__global__ __launch_bounds__(256, 5)
void k_tile_queue(const QueueCtl* __restrict__ qp, double* const* __restrict__ ptrs, int nbatch, QExtra extra, int ld,
                  unsigned long long* __restrict__ stats_)
{
    unsigned long long* const stats = STATS ? stats_ : nullptr;     // (the counters cost ~20 VGPRs: a kernel of their own)
    // (the control block by reference: by value its sixteen pointers stay in SGPRs across the contraction, the kernel
    // runs out of them and the spills cost VGPRs)
    const QueueCtl& q = *qp;
    // the two operand stages, 2 * 16 * (64 + 64 + 32) doubles, as DYNAMIC shared memory: with a static array the
    // compiler sees that LDS allows three workgroups per CU and spends the registers of three waves per SIMD (150+ VGPRs)
    // whatever the launch bound says
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned s_entry;
    // ---- placement (experiments; off unless GPRN_QUEUE_WG_PER_CU > 0).  The hardware decides where workgroups go;
    // which of them STAY can be decided here: every workgroup registers on its CU (HW_ID / XCC_ID); the first
    // `reserve_per_xcc` CUs of each XCD to be registered are vacated, everywhere else at most `per_cu` workgroups stay.
    // Measured: vacated CUs do not help the chain -- its launches are dealt to a shader engine and wait for THAT one to
    // have room (milliseconds, with every other CU of it full) instead of going to the empty CUs elsewhere.
    if (q.per_cu > 0) {
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            const unsigned key = xcc * 128u + ((hw >> 13) & 3u) * 32u + ((hw >> 12) & 1u) * 16u + ((hw >> 8) & 15u);
            unsigned* const reg = q.ctr + QC_COUNT * GPRN_QCTR_STRIDE + key;
            const unsigned n = __hip_atomic_fetch_add(reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
            unsigned state;
            if (n == 0) {
                const unsigned r = __hip_atomic_fetch_add(q.ctr + QC_XCC * GPRN_QCTR_STRIDE + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                state = r < (unsigned)q.reserve_per_xcc ? 0x30000u : 0x10000u;
                __hip_atomic_fetch_or(reg, state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                state = q_load(reg);
                for (int spin = 0; spin < 200 && !(state & 0x10000u); ++spin) { __builtin_amdgcn_s_sleep(8); state = q_load(reg); }
            }
            s_entry = (state & 0x20000u) ? 0u : (n < (unsigned)q.per_cu ? 1u : 0u);
        }
        __syncthreads();
        if (!s_entry) return;
        __syncthreads();
    }
    unsigned long long st_idle = 0, st_claim = 0, st_acq = 0, st_desc = 0, st_run = 0, st_done = 0, st_n = 0, st_k = 0, tq = 0, t_found = 0, t_start = 0;
#define Q_STAMP(acc) do { if (stats) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); acc += now_ - tq; tq = now_; } } while (0)
    if (stats) tq = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (threadIdx.x < 64) {
            unsigned long long t_pass = 0;
            const unsigned e = q_claim(q, blockIdx.x, stats ? &t_pass : nullptr);
            if (stats) { st_idle += t_pass - tq; tq = t_pass; }
            if (STATS && q.trace) t_found = __builtin_amdgcn_s_memrealtime();
            Q_STAMP(st_claim);
            if (e != GPRN_Q_EMPTY) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (threadIdx.x == 0) s_entry = e;
        }
        __syncthreads();
        Q_STAMP(st_acq);
        const unsigned e = __builtin_amdgcn_readfirstlane(s_entry);   // in an SGPR: the node, its pointers and modes stay scalar
        if (e == GPRN_Q_EMPTY) break;
        const unsigned m = e >> 24, sub = (e >> 21) & 7u, op = e & 0x1fffffu;
        const QOp* o = q.ops + op;
        const TileTask t = o->t;
        const int kind = o->kind, flags = o->flags;
        double* gp[GPRN_NBUF];
#pragma unroll
        for (int b2 = 0; b2 < GPRN_NBUF; ++b2)
            gp[b2] = m < (unsigned)nbatch ? ptrs[(size_t)m * GPRN_NBUF + b2] : extra.p[(m - nbatch) & (GPRN_Q_EXTRA - 1)][b2];
        if (stats) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the node and its pointers have arrived)
        Q_STAMP(st_desc);
        if (STATS && q.trace) t_start = __builtin_amdgcn_s_memrealtime();
        run_node(lds, t, kind, flags, sub, gp[0], gp[1], gp[2], gp[3], ld);
        Q_STAMP(st_run);
        q_complete(q, m, op, true);                        // (its barrier also covers s_entry and the LDS stages)
        Q_STAMP(st_done);
        if (STATS && q.trace && threadIdx.x == 0)
            q_trace(q, (unsigned long long)e | ((unsigned long long)blockIdx.x << 32), t_found, t_start, __builtin_amdgcn_s_memrealtime());
        st_n += 1; st_k += (unsigned)t.klen / 16 * (sub == GPRN_Q_WHOLE ? 4 : 1);
    }
#undef Q_STAMP
    if (stats && threadIdx.x == 0) {
        unsigned long long* o = stats + (size_t)blockIdx.x * 16;
        o[0] += st_idle; o[1] += st_acq; o[2] += st_run; o[3] += st_done; o[4] += st_n; o[5] += st_k; o[6] += 1; o[7] += st_claim;
        o[8] += st_desc;
    }
}

// ------------------------------------------------------------------ host side of a call
int queue_enabled(gprn_ctx* c)
{
    // opt-in (GPRN_QUEUE=1 or gprn_set_option "queue"): correct everywhere it was tried, but 45 % slower than the launch
    // schedule at BASELINE config 3 -- DESIGN.md §8 has the measurements and what they say
    if (c->queue_mode < 0) c->queue_mode = env_int("GPRN_QUEUE", 0) ? 1 : 0;
    return c->queue_mode == 1 && factor_use_flags(c) == 1;
}

int queue_check_waits(gprn_ctx* c)
{
    if (!c->d_qctr) return GPRN_OK;
    unsigned flag = 0;
    unsigned* const tmo = c->d_qctr + QC_TIMEOUT * GPRN_QCTR_STRIDE;
    HIP_TRY(c, hipMemcpy(&flag, tmo, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (flag) {
        if (env_int("GPRN_QUEUE_DEBUG", 0)) {
            // what the call looked like when it gave up: the counters and the CU registry
            std::vector<unsigned> h(GPRN_QCTR_WORDS);
            if (hipMemcpy(h.data(), c->d_qctr, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[gprn] queue at time-out: left %u bell %u;", h[QC_LEFT * GPRN_QCTR_STRIDE], h[QC_BELL * GPRN_QCTR_STRIDE]);
                for (int cl = 0; cl < GPRN_QCLASSES; ++cl)
                    fprintf(stderr, " class %d head %u tail %u;", cl, h[(QC_HEAD + cl) * GPRN_QCTR_STRIDE], h[(QC_TAIL + cl) * GPRN_QCTR_STRIDE]);
                fprintf(stderr, "\n[gprn] CUs registered per XCD:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %u", h[QC_XCC * GPRN_QCTR_STRIDE + x]);
                int hist[8] = {0}, reserved = 0, cus = 0;
                for (int k = 0; k < GPRN_QCU_WORDS; ++k) {
                    const unsigned w = h[QC_COUNT * GPRN_QCTR_STRIDE + k];
                    if (!w) continue;
                    ++cus;
                    if (w & 0x20000u) ++reserved;
                    hist[std::min(7u, w & 0xffffu)] += 1;
                }
                fprintf(stderr, "; %d CUs seen, %d reserved; CUs by workgroups registered:", cus, reserved);
                for (int i = 0; i < 8; ++i) fprintf(stderr, " %d:%d", i, hist[i]);
                fprintf(stderr, "\n");
            }
        }
        hipMemset(tmo, 0, sizeof(unsigned));
        c->err = "factorisation: a wait of the dataflow schedule timed out";
        return GPRN_E_WAIT_TIMEOUT;
    }
    return GPRN_OK;
}

template <typename T>
static int grow(gprn_ctx* c, T** p, size_t* cap, size_t want)
{
    if (*cap >= want && *p) return GPRN_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    if (hipMalloc((void**)p, want * sizeof(T)) != hipSuccess) { c->err = "queue: device allocation failed"; return GPRN_E_NOMEM; }
    *cap = want;
    return GPRN_OK;
}

int factor_invert_queue(gprn_ctx* c, int nbatch, int set)
{
    const int T = c->T, ld = c->ld;
    const int outer = c->outers[set].empty() ? GPRN_OUTER : c->outers[set][0].k1 - c->outers[set][0].k0;
    QueuePlanRef*& P = c->qplan[set];
    if (P && (P->T != T || P->outer != outer)) { plan_free(P); P = nullptr; }
    if (!P) {
        P = build_plan(c, T, ld, outer);
        if (!P) return GPRN_E_NOMEM;
    }
    if (nbatch > 200) { c->err = "queue schedule: batch too large"; return GPRN_E_ARG; }
    // the previous phase's X^T X (run_phase, api.hip) rides along as filler
    const int n_extra = c->q_lauum.n;
    const int nmat = nbatch + n_extra;
    const uint32_t nops = (uint32_t)P->ops.size();
    if (nops >= (1u << 21)) { c->err = "queue schedule: too many nodes"; return GPRN_E_ARG; }
    // ---- buffers
    int rc;
    if ((rc = grow(c, &c->d_qstate, &c->qstate_cap, (size_t)nmat * nops))) return rc;
    size_t cap[GPRN_QCLASSES], total = 0;
    for (int cl = 0; cl < GPRN_QCLASSES; ++cl) {
        cap[cl] = (size_t)nbatch * P->ent_main[cl] + (cl == GPRN_QCLASSES - 1 ? (size_t)n_extra * P->ent_lauum : 0);
        total += cap[cl] + 32;
    }
    if ((rc = grow(c, &c->d_qslots, &c->qslots_cap, total))) return rc;
    if (!c->d_qctr) {
        // counters, the time-out word and its budget, then an image of the control block for the workers
        HIP_TRY(c, hipMalloc(&c->d_qctr, GPRN_QCTR_WORDS * sizeof(unsigned) + sizeof(QueueCtl)));
        HIP_TRY(c, hipMemset(c->d_qctr, 0, GPRN_QCTR_WORDS * sizeof(unsigned) + sizeof(QueueCtl)));
        c->q_budget_ms = -1;
    }
    if (!c->ev_qreset) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_qreset, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_qdone, hipEventDisableTiming));
    }
    unsigned* const tmo = c->d_qctr + QC_TIMEOUT * GPRN_QCTR_STRIDE;
    if (c->q_budget_ms != c->wait_budget_ms) {
        const unsigned ticks = (unsigned)std::min<long long>(0xffffffffll, (long long)c->wait_budget_ms * 100000ll);
        HIP_TRY(c, hipMemcpy(tmo + 1, &ticks, sizeof(unsigned), hipMemcpyHostToDevice));
        c->q_budget_ms = c->wait_budget_ms;
    }
    static int want_stats = -1;
    if (want_stats < 0) want_stats = env_int("GPRN_QUEUE_STATS", 0);
    if (want_stats && !c->d_qstats) {
        HIP_TRY(c, hipMalloc(&c->d_qstats, (size_t)4096 * 16 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_qstats, 0, (size_t)4096 * 16 * sizeof(unsigned long long)));
    }
    static int want_trace = -1;
    if (want_trace < 0) want_trace = env_int("GPRN_QUEUE_TRACE", 0);
    if (want_trace > 0 && !c->d_qtrace) {
        HIP_TRY(c, hipMalloc(&c->d_qtrace, ((size_t)want_trace * 4 + 1) * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_qtrace, 0, ((size_t)want_trace * 4 + 1) * sizeof(unsigned long long)));
        c->qtrace_cap = want_trace;
    }
    // ---- the rows of the X^T X matrices travel as a kernel argument
    QExtra extra;
    memset(&extra, 0, sizeof(extra));
    if (n_extra > GPRN_Q_EXTRA || (size_t)n_extra * GPRN_NBUF > c->q_lauum.rows.size()) {
        c->err = "queue schedule: too many X^T X matrices handed over";
        return GPRN_E_ARG;
    }
    for (int e = 0; e < n_extra; ++e)
        for (int b2 = 0; b2 < GPRN_NBUF; ++b2) extra.p[e][b2] = c->q_lauum.rows[(size_t)e * GPRN_NBUF + b2];
    QueueCtl q;
    q.ops = P->d_ops; q.succ = P->d_succ; q.state = c->d_qstate;
    {
        unsigned* at = c->d_qslots;
        for (int cl = 0; cl < GPRN_QCLASSES; ++cl) { q.slots[cl] = at; q.cap[cl] = (unsigned)cap[cl]; at += cap[cl] + 32; }
    }
    q.ctr = c->d_qctr; q.timed_out = tmo; q.nops = (int)nops;
    q.trace = c->d_qtrace; q.trace_cap = c->qtrace_cap; q.call_id = c->q_calls++;
    q.per_cu = env_int("GPRN_QUEUE_WG_PER_CU", 0);                 // > 0: placement by registration (experiments)
    q.reserve_per_xcc = env_int("GPRN_QUEUE_RESERVE", 0);
    size_t left = 0;
    for (int cl = 0; cl < GPRN_QCLASSES; ++cl) left += cap[cl];
    hipStream_t s0 = c->stream, s1 = c->stream3;
    // test hook (gprn_set_option "withhold_inner"): a tile node in the middle of matrix 0's graph never becomes ready, so
    // the chain's wait for it gives up after the budget and the call is re-run on HIP events
    int hold_op = -1;
    if (c->withhold_inner > 0)
        for (uint32_t v = P->n_main / 2; v < P->n_main && hold_op < 0; ++v)
            if (P->ops[v].kind != QK_CHAIN) hold_op = (int)v;
    // ---- reset on the chain stream, then the workers on the side stream
    QueueCtl* const qimg = (QueueCtl*)(c->d_qctr + GPRN_QCTR_WORDS);
    hipLaunchKernelGGL(k_queue_init, dim3(512), dim3(256), 0, s0, q, qimg, (const uint32_t*)P->d_init, (int)P->n_main, nbatch,
                       (int)P->lauum0, (int)P->n_lauum, n_extra, (unsigned)left, hold_op);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_qreset, s0));
    HIP_TRY(c, hipStreamWaitEvent(s1, c->ev_qreset, 0));
    static int n_cu = 0;
    if (!n_cu) { hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device); if (n_cu <= 0) n_cu = 256; }
    // TWO workgroups per CU, on every CU: at 96 VGPRs a worker wave leaves a diagonal-block workgroup (one wave per SIMD
    // with 282 VGPRs, 46.6 KB of LDS) room on the same CU, so the chain's kernels can be dispatched ANYWHERE.  That matters:
    // the dispatcher does not search the chip for a CU with room -- with CUs set aside for the chain (placement, below)
    // the chain's launches waited milliseconds for the shader engine they had been dealt to, while the reserved CUs
    // elsewhere stood empty.  An LDS pad keeps a third worker off a CU (2 x 54.9 + 46.6 KB fit the 160).
    const size_t static_lds = 2 * 16 * (64 + 64 + 32) * sizeof(double) + 64;      // (the stages are dynamic LDS: k_tile_queue)
    const int wg_per_cu = std::min(3, std::max(1, env_int("GPRN_QUEUE_LAUNCH_PER_CU", 2)));
    const size_t share = lds_limit(c->device) / (wg_per_cu + 1) + 1024;            // more than a (wg_per_cu + 1)-th of the CU's LDS
    const size_t pad = wg_per_cu < 3 && share > static_lds ? share - static_lds : 0;
    const int nwg = std::max(1, env_int("GPRN_QUEUE_WORKERS", wg_per_cu * n_cu));
    if (nwg > 4096) { c->err = "queue schedule: too many workers"; return GPRN_E_ARG; }
    prof_begin(c, GPRN_T_UPDATE, s1);
    if (c->d_qstats || c->d_qtrace)
        hipLaunchKernelGGL(k_tile_queue<true>, dim3(nwg), dim3(256), static_lds + pad, s1, (const QueueCtl*)qimg, (double* const*)c->d_ptrs, nbatch, extra, ld, c->d_qstats);
    else
        hipLaunchKernelGGL(k_tile_queue<false>, dim3(nwg), dim3(256), static_lds + pad, s1, (const QueueCtl*)qimg, (double* const*)c->d_ptrs, nbatch, extra, ld, c->d_qstats);
    prof_end(c);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_qdone, s1));
    if (c->chain_started) {
        // what the caller handed over for "beside this factorisation" (run_phase, api.hip: the previous phase's X^T X
        // and the traces of quirk Q1 that read it): the product itself went into the queue above, the rest follows
        // the worker kernel on the bulk stream
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_qdone, 0));
        c->q_lauum_in_queue = n_extra > 0;
        std::function<int()> f;
        f.swap(c->chain_started);
        rc = f();
        c->q_lauum_in_queue = false;
        if (rc) return rc;
    }
    // ---- the chain: three launches per tile step, each polling its own node
    static int spin_max = -1;
    if (spin_max < 0) spin_max = env_int("GPRN_SPIN_MAX_BATCH", 2);
    for (int k = 0; k < T; ++k) {
        if ((rc = launch_diag_q(c, c->d_ptrs, nbatch, ld, k, c->d_info_cur, s0, q, P->diag_op[k]))) return rc;
        if (k + 1 == T) break;
        const bool wait_kernel = nbatch > spin_max;
        if (wait_kernel) {
            hipLaunchKernelGGL(k_queue_wait, dim3(1), dim3(64), 0, s0, q, P->l_op[k], nbatch, P->u_op[k], 1u);
            HIP_TRY(c, hipGetLastError());
        }
        if ((rc = launch_tile_rows_q(c, k, c->d_ptrs, nbatch, ld, 0, s0, q, P->l_op[k], wait_kernel))) return rc;
        if ((rc = launch_tile_rows_q(c, k, c->d_ptrs, nbatch, ld, 1, s0, q, P->u_op[k], wait_kernel))) return rc;
    }
    HIP_TRY(c, hipStreamWaitEvent(s0, c->ev_qdone, 0));
    c->q_lauum.n = 0;
    return GPRN_OK;
}

// Diagnostic (tests, probes): a list of INDEPENDENT tile tasks through the worker kernel -- every node ready from the
// start, no chain -- timed with HIP events.  d_ptrs: one row of GPRN_NBUF pointers.  whole: one queue entry per node
// instead of one per 64 x 64 quarter.
int queue_run_independent(gprn_ctx* c, const std::vector<TileTask>& tasks, double** d_ptrs, int ld, bool whole, int reps, float* ms)
{
    const uint32_t n = (uint32_t)tasks.size();
    std::vector<QOp> ops(n);
    std::vector<uint32_t> init(n);
    for (uint32_t i = 0; i < n; ++i) {
        memset(&ops[i], 0, sizeof(QOp));
        ops[i].t = tasks[i];
        ops[i].kind = QK_TILE;
        ops[i].cls = GPRN_QCLASSES - 1;
        ops[i].nent = whole ? 1 : 4;
        init[i] = (uint32_t)ops[i].nent << 16;
    }
    QOp* d_ops = nullptr;
    uint32_t* d_init = nullptr;
    unsigned* d_slots = nullptr;
    const size_t nslots = (size_t)n * (whole ? 1 : 4);
    HIP_TRY(c, hipMalloc(&d_ops, n * sizeof(QOp)));
    HIP_TRY(c, hipMalloc(&d_init, n * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&d_slots, (nslots + 64) * sizeof(unsigned)));
    HIP_TRY(c, hipMemcpy(d_ops, ops.data(), n * sizeof(QOp), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_init, init.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    std::vector<unsigned> hs(nslots);
    for (uint32_t i = 0; i < n; ++i)
        for (unsigned e = 0; e < (whole ? 1u : 4u); ++e) hs[(size_t)i * (whole ? 1 : 4) + e] = q_entry(0, whole ? GPRN_Q_WHOLE : e, i);
    int rc;
    if ((rc = grow(c, &c->d_qstate, &c->qstate_cap, (size_t)n))) return rc;
    if (!c->d_qctr) {
        HIP_TRY(c, hipMalloc(&c->d_qctr, GPRN_QCTR_WORDS * sizeof(unsigned) + sizeof(QueueCtl)));
        HIP_TRY(c, hipMemset(c->d_qctr, 0, GPRN_QCTR_WORDS * sizeof(unsigned) + sizeof(QueueCtl)));
        c->q_budget_ms = -1;
    }
    unsigned* const tmo = c->d_qctr + QC_TIMEOUT * GPRN_QCTR_STRIDE;
    const unsigned ticks = 200000000u;
    HIP_TRY(c, hipMemcpy(tmo + 1, &ticks, sizeof(unsigned), hipMemcpyHostToDevice));
    c->q_budget_ms = -1;
    QueueCtl q;
    memset(&q, 0, sizeof(q));
    q.ops = d_ops; q.succ = d_init /* unused: no successors */; q.state = c->d_qstate;
    for (int cl = 0; cl < GPRN_QCLASSES; ++cl) { q.slots[cl] = d_slots; q.cap[cl] = cl == GPRN_QCLASSES - 1 ? (unsigned)nslots : 0u; }
    q.ctr = c->d_qctr; q.timed_out = tmo; q.nops = (int)n;
    q.per_cu = env_int("GPRN_QUEUE_WG_PER_CU", 0); q.reserve_per_xcc = 0;
    QueueCtl* const qimg = (QueueCtl*)(c->d_qctr + GPRN_QCTR_WORDS);
    HIP_TRY(c, hipMemcpy(qimg, &q, sizeof(q), hipMemcpyHostToDevice));
    QExtra extra;
    memset(&extra, 0, sizeof(extra));
    static int n_cu = 0;
    if (!n_cu) { hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device); if (n_cu <= 0) n_cu = 256; }
    const size_t static_lds = 2 * 16 * (64 + 64 + 32) * sizeof(double) + 64;
    const int wg_per_cu = std::min(3, std::max(1, env_int("GPRN_QUEUE_LAUNCH_PER_CU", 2)));
    const size_t share = lds_limit(c->device) / (wg_per_cu + 1) + 1024;
    const size_t pad = wg_per_cu < 3 && share > static_lds ? share - static_lds : 0;
    const int nwg = std::max(1, env_int("GPRN_QUEUE_WORKERS", wg_per_cu * n_cu));
    if (env_int("GPRN_QUEUE_STATS", 0) && !c->d_qstats) {
        HIP_TRY(c, hipMalloc(&c->d_qstats, (size_t)4096 * 16 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_qstats, 0, (size_t)4096 * 16 * sizeof(unsigned long long)));
    }
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    float total = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        HIP_TRY(c, hipMemcpy(c->d_qstate, init.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(d_slots, hs.data(), nslots * sizeof(unsigned), hipMemcpyHostToDevice));
        std::vector<unsigned> ctr(GPRN_QCTR_WORDS, 0u);
        ctr[(QC_TAIL + GPRN_QCLASSES - 1) * GPRN_QCTR_STRIDE] = (unsigned)nslots;      // tail of the last class
        ctr[QC_LEFT * GPRN_QCTR_STRIDE] = (unsigned)nslots;
        ctr[QC_TIMEOUT * GPRN_QCTR_STRIDE + 1] = ticks;
        HIP_TRY(c, hipMemcpy(c->d_qctr, ctr.data(), ctr.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        if (c->d_qstats)
            hipLaunchKernelGGL(k_tile_queue<true>, dim3(nwg), dim3(256), static_lds + pad, c->stream, (const QueueCtl*)qimg,
                               (double* const*)d_ptrs, 1, extra, ld, c->d_qstats);
        else
            hipLaunchKernelGGL(k_tile_queue<false>, dim3(nwg), dim3(256), static_lds + pad, c->stream, (const QueueCtl*)qimg,
                               (double* const*)d_ptrs, 1, extra, ld, c->d_qstats);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipEventSynchronize(e1));
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, e0, e1));
        if (r) total += t;
    }
    *ms = total / reps;
    queue_print_stats(c);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d_ops); (void)hipFree(d_init); (void)hipFree(d_slots);
    return queue_check_waits(c);
}

// Host-only view of the graph for tests (tests/test_queue_plan.py): nodes and edges of the plan for T tile steps and
// outer panels of `outer` tiles.  ops_out: n x 12 int64 per node: kind, class, entries, flags, c_buf, a_buf, b_buf, modes,
// c_off, a_off, b_off, klen (ld = 128 T; chain nodes: c_buf = 0 diag / 1 L / 2 update, klen = tile step); edges_out:
// m x 2 (from, to).  Null pointers: counts only.  No GPU is touched.
extern "C" int gprn_test_queue_plan(int T, int outer, int64_t* n_ops, int64_t* n_edges, int64_t* ops_out, int64_t* edges_out)
{
    if (T < 1 || outer < 1 || !n_ops || !n_edges) return GPRN_E_ARG;
    QueuePlanRef* P = build_plan_host(T, T * GPRN_TILE, outer);
    *n_ops = P->n_main;
    *n_edges = (int64_t)P->succ.size();
    if (ops_out)
        for (uint32_t v = 0; v < P->n_main; ++v) {
            const QOp& o = P->ops[v];
            int64_t* r = ops_out + (size_t)v * 12;
            r[0] = o.kind; r[1] = o.cls; r[2] = o.nent; r[3] = o.flags;
            r[4] = o.t.c_buf; r[5] = o.t.a_buf; r[6] = o.t.b_buf; r[7] = o.t.modes;
            r[8] = o.t.c_off; r[9] = o.t.a_off; r[10] = o.t.b_off; r[11] = o.t.klen;
        }
    if (edges_out) {
        size_t e = 0;
        for (uint32_t v = 0; v < P->n_main; ++v)
            for (uint32_t i = 0; i < P->ops[v].nsucc; ++i) {
                edges_out[2 * e] = v;
                edges_out[2 * e + 1] = P->succ[P->ops[v].succ0 + i];
                ++e;
            }
    }
    delete P;
    return GPRN_OK;
}
