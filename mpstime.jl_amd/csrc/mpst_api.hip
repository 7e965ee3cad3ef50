// C ABI of libmpstime_hip.so (include/mpstime_hip.h): context management, host<->device
// marshalling, the sweep driver (src/Training/RealRealHighDimension.jl:724-851 restated as a
// stream of kernel launches with no host read-back inside a sweep) and the RCCL plumbing.
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include "mpst_internal.h"

using namespace mpst;

namespace {

thread_local std::string g_err;  // errors raised without a context (mpst_create)

// ---- RCCL, bound at run time ----------------------------------------------------------------------------------------------
// The library is NOT linked against librccl: a host process that has imported torch already holds torch's own copy
// (torch/lib/librccl.so, same SONAME librccl.so.1 as /opt/rocm/lib's), and two copies of a collective library in one process
// - two sets of bootstrap threads, IPC registries and topology caches - is a hazard on first multi-GPU contact.  Order:
// MPST_RCCL_LIB (explicit path), then whatever librccl.so.1 the process has ALREADY loaded (RTLD_NOLOAD: the host's copy
// wins), then the system one.  mpst_comm_init compares the version of the bound library with the header this file was
// compiled against (same major version, see there) and mpst_comm_library reports path and version.
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    int version = 0;
    std::string path, how, err;
};
void rccl_load(Rccl& r) {
    const char* envp = getenv("MPST_RCCL_LIB");
    if (envp && !*envp) envp = nullptr;                // an empty value means unset
    if (envp) {
        r.h = dlopen(envp, RTLD_NOW | RTLD_LOCAL);
        r.how = "MPST_RCCL_LIB";
    }
    if (!r.h && !envp) {
        for (const char* nm : {"librccl.so.1", "librccl.so"}) {
            if ((r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) { r.how = "already loaded by the host process"; break; }
        }
    }
    if (!r.h && !envp) {
        for (const char* nm : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) {
            (void)dlerror();                           // the message reported below belongs to the last attempt, not to the probes above
            if ((r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) { r.how = "system library"; break; }
        }
    }
    if (!r.h) {
        const char* de = dlerror();
        r.err = std::string("librccl could not be loaded: ") + (de ? de : "not found");
        return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString || !r.GetVersion) {
        r.err = "librccl lacks a required entry point";
        r.h = nullptr;
        return;
    }
    Dl_info di;
    if (dladdr((void*)r.AllReduce, &di) && di.dli_fname) r.path = di.dli_fname;
    (void)r.GetVersion(&r.version);
}
// one-time, thread-safe: contexts on different threads may call mpst_comm_init / mpst_comm_library concurrently
Rccl* rccl_get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return &r;
}
// the entry points of the bound library, or null with the reason in *why
Rccl* rccl_ready(std::string* why) {
    Rccl* r = rccl_get();
    if (!r->h) {
        if (why) *why = r->err;
        return nullptr;
    }
    return r;
}

enum KClass { K_YHAT = 0, K_GRAD, K_UPDATE, K_GRAM, K_EIG_TRI, K_SPLIT, K_ENV, K_BT, K_ALLREDUCE, K_EIG_VEC, K_EIG_FIN, K_NCLASS };

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    mpst_options opt{};
    bool have_opt = false;
    int T = 0, d = 0, C = 0;
    int cap = 0;              // capacity bond dimension of all device buffers
    // element type of the data sets and of the MPS (mpst_set_dataset's dtype = opts.dtype, RealRealHighDimension.jl:442).
    // typed: everything runs through the element-typed kernels of mpst_typed.hip (always for fp32 / complex; MPST_TYPED=1
    // sends Float64 through them too - the cross-check of the two implementations); the buffers marked (E) below then hold
    // elements of esz bytes behind their double* names.
    int dtype = MPST_F64;
    bool have_dtype = false, typed = false;
    int zw = 1;               // 2: complex
    size_t esz = 8;           // bytes per element
    double* tnorm_scratch = nullptr;   // typed normalize: three complex cap x cap matrices
    int32_t *xLE = nullptr, *xRE = nullptr, *yexp = nullptr;      // typed: binary exponents of the environment rows / overlaps
    int32_t *xchainL[2] = {nullptr, nullptr}, *xchainR[2] = {nullptr, nullptr};
    DataSet ds[2];
    // MPS
    bool have_mps = false;
    double* sites = nullptr;
    int64_t site_stride = 0;
    int32_t* chi = nullptr;         // device [T+1]
    int32_t* label_site = nullptr;  // device
    // caches + workspaces (train set)
    double *LE = nullptr, *RE = nullptr;
    int64_t cache_elems = 0;
    double *bt = nullptr, *yhat = nullptr, *tile_loss = nullptr, *partial = nullptr, *gradbuf = nullptr;
    double *gram = nullptr, *lam = nullptr, *E = nullptr, *eig_ws = nullptr;
    double *btn = nullptr, *norm_part = nullptr;
    // four-launch chain (k_grad_s, k_gram_upd, k_eig_trivec, k_bond_tail): bt_new once more as [c][y][x] for the tail's contraction going
    // right; which bond's overlaps the last tail launch left in b2_ypart (valid while nothing else touched the context: bond_seq / epoch);
    // a tail whose on-device verification failed marks the sweep (DevScalars::redo) and the rest of it is redone on the six-launch chain
    double* btnT = nullptr;
    unsigned long long* tail_span = nullptr;    // diagnostics: (start, end) stamps of every workgroup of the stamped k_bond_tail launch
    bool chain4_ok = false, chain4_hold = false;
    int ynext_lid = -1;
    uint64_t ynext_epoch = 0, ynext_seq = 0, bond_seq = 0;
    int tail_redos = 0;
    int32_t ss_counts[2] = {0, 0};      // subspace eigensolver: bonds attempted / accepted, read where a sweep or a bond step synchronises anyway
    int tail_force_fail = -1, tail_launches = 0;      // test hook (MPST_TAIL_FORCE_REDO=n): the n-th tail launch of the context reports a failed verification
    // sliced bond GEMMs (k_yhat_s / k_grad_s): slice contributions to yhat, loss pieces, arrival tickets
    double *b2_ypart = nullptr, *b2_lossp = nullptr;
    unsigned int* b2_tick = nullptr;
    unsigned long long* b2_dbg = nullptr;   // bring-up stamps of the sliced kernels (MPST_B2_DEBUG builds)
    int b2_ksplit = 0, b2_norm_parts = 0;
    bool b2 = false;            // the fused chain uses k_yhat_s + k_grad_s instead of k_bond_fused + k_fused_reduce
    double* loss_trace = nullptr;   // track_cost: [2(T-1)][update_iters + 1]
    int n_norm_part = 0;
    bool fused = false;        // bond tensors <= MAX_DIM^2 and no rescale[1]: the 7-launch chain of mpst_fused.hip
    int64_t partial_elems = 0;
    DevScalars* sc = nullptr;
    // eval scratch
    double *chainL[2] = {nullptr, nullptr}, *chainR[2] = {nullptr, nullptr}, *yeval = nullptr, *out3 = nullptr;
    int64_t* conf = nullptr;
    int32_t* pred = nullptr;
    int64_t eval_N = 0;
    double* norm2 = nullptr;
    double* norm_scratch = nullptr;   // 3*cap*cap doubles for k_norm2 when they exceed its LDS
    BigEig* big = nullptr;            // d*cap > MAX_DIM: library eigensolver at the capacity size (fallback of the blocked one)
    BlockedEig* blk = nullptr;        // d*cap > MAX_DIM: hand-written blocked eigensolver
    int64_t big_fallbacks = 0;        // bonds on which the blocked solver's verification asked for the library
    // large bonds, sweeps: the verdict of the blocked eigensolver is read once per sweep instead of once per bond (no host
    // synchronisation inside the sweep); a sweep in which any bond failed is redone from a snapshot, bond by bond
    bool big_opt = false, big_opt_active = false;
    double* snap_sites = nullptr;
    int32_t* snap_chi = nullptr;      // [T + 2]: chi, label_site
    DevScalars* snap_sc = nullptr;
    int big_redos = 0, big_force_fail = -1, big_solves = 0;
    int big_cooldown = 0;             // sweeps left that read the verdict per bond after a sweep had to be redone
    bool ws_ready = false;      // training workspace (caches, bond tensor, gradient, eigensolver) allocated for the current sizes
    bool eval_ready = false;    // evaluation scratch (chains, yeval, pred) allocated for max(N_train, N_test)
    bool caches_valid = false;  // LE / RE describe the current MPS: set by mpst_build_caches, cleared by whatever invalidates them
    int host_label_site = -1;   // host mirror of *label_site (set_mps, bond_step and sweep move it deterministically)
    // multi-GPU
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    // MPST_FORCE_COLLECTIVE=1 at mpst_comm_init: the sharded launch chain (loss in the message, all-reduce, norm pieces from the
    // summed gradient, plain stream) also with ONE rank - the RCCL leg then runs on every 1-GPU CI box (tests/test_gpu_multi.py)
    bool force_coll = false;
    // one-shot direct-write all-reduce (mpst_allreduce.hip): this rank's inbox (fine-grained device memory, exported
    // over IPC) and the peers' inboxes as mapped here
    void* ipc_local = nullptr;           // [2][nranks][slot] doubles | 16 flags | 2 counters
    void* ipc_peer[AR_MAX_RANKS] = {nullptr};
    int64_t ipc_slot = 0;
    size_t ipc_flag_off = 0, ipc_ctr_off = 0, ipc_bytes = 0;
    bool use_ipc = false;                // all peers attached: gradient and evaluation sums go through the one-shot path
    bool ipc_dead = false;               // a one-shot all-reduce timed out: its cross-rank state is undefined until the inboxes are exported again
    unsigned long long ar_epoch = 0;
    // profiling
    unsigned prof_mask = 0;
    std::vector<hipEvent_t> ev_pool;
    struct Rec { int k; size_t e0, e1; };
    std::vector<Rec> recs;
    size_t ev_used = 0;
    double prof_us[16] = {0};
    double impute_phase_s[2] = {0, 0};   // last imputation call: environment pass, density sweep
    int impute_batched = 0;              // ... and whether its sweep ran sixteen instances per workgroup (k_imp_leftb)
    int impute_trig = 0;                 // ... and whether its densities were evaluated in closed form (Fourier states on a uniform grid)
    int64_t prof_cnt[16] = {0};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // one full sweep captured as a hipGraph (bond dimensions live on the device and every grid is sized
    // for the capacity, so the launch sequence of a sweep never changes between sweeps); `epoch` is
    // bumped by every call that changes what the captured kernels were given
    hipGraphExec_t sweep_graph = nullptr;
    uint64_t epoch = 1, graph_epoch = 0;
    // mpst_sweep_batch with this context as the lead: the Views of the K fits on the device ([2][K]: plain, and the Gram launch's
    // variant), the captured sweep and what it was captured for (context pointers and their epochs)
    int batch_hint = 1;            // mpst_set_batch_hint: this context will be advanced in batches of about that many fits
    View* batch_views = nullptr;
    int batch_cap = 0;
    hipGraphExec_t batch_graph = nullptr;
    std::vector<std::pair<uint64_t, uint64_t>> batch_key;   // (uid, epoch) per member: an address can be handed out again, a uid cannot
    uint64_t uid = 0;              // process-wide, never reused (mpst_create)
};

int fail(Ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_err = buf;
    return code;
}

#define HIPC(c, call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), MPST_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

template <typename T>
int dalloc(Ctx* c, T** p, int64_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (n <= 0) n = 1;
    hipError_t e = hipMalloc((void**)p, (size_t)n * sizeof(T));
    if (e != hipSuccess) return fail(c, MPST_ERR_NOMEM, "hipMalloc of %lld bytes failed: %s", (long long)(n * sizeof(T)), hipGetErrorString(e));
    return 0;
}
template <typename T>
void dfree(T** p) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
}

// (E) buffers: `n` elements of c->esz bytes behind a double* name
int dalloc_e(struct Ctx* c, double** p, int64_t n);

void free_dataset(DataSet& s) {
    dfree(&s.phi); dfree(&s.label); dfree(&s.tiles); dfree(&s.chunks); dfree(&s.cls_chunk_off); dfree(&s.cls_off); dfree(&s.inv_count);
    for (int k = 0; k < 2; ++k) { dfree(&s.parts[k]); dfree(&s.part_off[k]); }
    s = DataSet();
}

View make_view(Ctx* c, int which) {
    View v{};
    const DataSet& s = c->ds[which];
    v.T = c->T; v.d = c->d; v.C = c->C; v.chi_max = c->opt.chi_max; v.cap = c->cap;
    v.N = s.N;
    v.invN = s.Nglobal > 0 ? 1.0 / (double)s.Nglobal : 0.0;
    v.phi = s.phi; v.label = s.label; v.tiles = s.tiles; v.chunks = s.chunks;
    v.cls_chunk_off = s.cls_chunk_off; v.inv_count = s.inv_count;
    v.ntiles = s.ntiles; v.nchunks = s.nchunks;
    v.chi = c->chi; v.label_site = c->label_site; v.sites = c->sites; v.site_stride = c->site_stride;
    v.LE = c->LE; v.RE = c->RE; v.bt = c->bt; v.yhat = c->yhat; v.tile_loss = c->tile_loss;
    v.ss_bt = c->bt; v.ss_f32 = 0;
    v.partial = c->partial; v.gradbuf = c->gradbuf; v.gram = c->gram; v.lam = c->lam; v.E = c->E; v.eig_ws = c->eig_ws; v.sc = c->sc;
    v.loss = c->opt.loss; v.optimiser = c->opt.optimiser; v.rescale_before = c->opt.rescale_before;
    v.rescale_after = c->opt.rescale_after; v.train_sep = c->opt.train_classes_separately; v.svd_alg = c->opt.svd_alg;
    v.eta = c->opt.eta; v.cutoff = c->opt.cutoff;
    const int pk = c->opt.loss == MPST_LOSS_MSE ? 1 : 0;
    v.parts = s.parts[pk]; v.part_off = s.part_off[pk]; v.nparts = s.nparts[pk];
    v.norm_part = c->norm_part; v.n_norm_part = c->n_norm_part; v.btn = c->btn;
    v.btnT = (which == MPST_TRAIN && c->chain4_ok && !c->chain4_hold) ? c->btnT : nullptr;
    v.trace = nullptr; v.trace_it = 0; v.yhat_scaled = 0;
    v.cls_off = s.cls_off; v.ypart = c->b2_ypart; v.lossp = c->b2_lossp; v.tick = c->b2_tick; v.b2_ksplit = c->b2_ksplit; v.b2_nw = (c->batch_hint > 1 && c->d == 4) ? 4 : 8; v.dbg = c->b2_dbg;
    {
        int32_t so = 0, to = 0;
        for (int k = 0; k <= MAX_C; ++k) {
            v.kcls_off[k] = so;
            v.kcls_tile[k] = to;
            if (k < (int)s.counts.size()) {
                so += (int32_t)s.counts[k];
                to += (int32_t)((s.counts[k] + TILE_S - 1) / TILE_S);
            }
        }
    }
    return v;
}

int dalloc_e(Ctx* c, double** p, int64_t n) {
    uint8_t* q = (uint8_t*)*p;
    *p = nullptr;
    int rc = dalloc(c, &q, std::max<int64_t>(n, 1) * (int64_t)c->esz);
    *p = (double*)q;
    return rc;
}
inline double* eoff(Ctx* c, double* base, int64_t elems) { return (double*)((char*)base + (size_t)elems * c->esz); }

TView make_tview(Ctx* c, int which) {
    TView t{};
    const DataSet& s = c->ds[which];
    t.T = c->T; t.d = c->d; t.C = c->C; t.chi_max = c->opt.chi_max; t.cap = c->cap;
    t.cx = c->zw == 2; t.f32 = (c->dtype == MPST_F32 || c->dtype == MPST_C64);
    t.N = s.N;
    t.invN = s.Nglobal > 0 ? 1.0 / (double)s.Nglobal : 0.0;
    t.phi = s.phi; t.label = s.label; t.tiles = s.tiles; t.chunks = s.chunks; t.cls_chunk_off = s.cls_chunk_off; t.inv_count = s.inv_count;
    t.ntiles = s.ntiles; t.nchunks = s.nchunks;
    t.chi = c->chi; t.label_site = c->label_site; t.sites = c->sites; t.site_stride = c->site_stride;
    t.LE = c->LE; t.RE = c->RE; t.xLE = c->xLE; t.xRE = c->xRE; t.yexp = c->yexp;
    t.bt = c->bt; t.yhat = c->yhat; t.tile_loss = c->tile_loss; t.partial = c->partial;
    t.gradbuf = c->gradbuf; t.norm_part = c->norm_part; t.n_norm_part = c->n_norm_part;
    t.gram = c->gram; t.E = c->E; t.ldE = c->zw * c->cap; t.sc = c->sc;
    t.loss = c->opt.loss; t.optimiser = c->opt.optimiser; t.rescale_before = c->opt.rescale_before; t.rescale_after = c->opt.rescale_after;
    t.train_sep = c->opt.train_classes_separately;
    t.eta = c->opt.eta; t.cutoff = c->opt.cutoff;
    t.trace = nullptr; t.trace_it = 0; t.yhat_scaled = 0;
    return t;
}
// what the fp64 eigensolvers see of a typed context: the (embedded) Gram matrix, doubled counts for complex elements
View make_eig_view(Ctx* c) {
    View v{};
    v.T = c->T; v.d = c->d; v.C = c->C;
    v.chi_max = c->zw * c->opt.chi_max;
    v.cap = c->zw * c->cap;
    v.chi = c->chi; v.label_site = c->label_site;
    v.gram = c->gram; v.lam = c->lam; v.E = c->E; v.eig_ws = c->eig_ws; v.sc = c->sc;
    v.rescale_after = c->opt.rescale_after; v.svd_alg = c->opt.svd_alg; v.cutoff = c->opt.cutoff;
    v.zw = c->zw;
    v.ss_bt = c->bt;                                             // the subspace eigensolver reads the bond tensor itself
    v.ss_f32 = (c->dtype == MPST_F32 || c->dtype == MPST_C64) ? 1 : 0;
    return v;
}

inline bool multi(const Ctx* c) { return c->nranks > 1 || c->force_coll; }

void ipc_release(struct Ctx* c);

// (re)allocate the evaluation scratch: sized by the larger of the two data sets, independent of the training
// workspace, so that (re)loading a TEST set never touches the environment caches
int ensure_eval(Ctx* c) {
    if (c->eval_ready) return 0;
    int rc;
    const int64_t en = std::max<int64_t>(1, std::max(c->ds[0].N, c->ds[1].N));
    c->eval_N = en;
    for (int k = 0; k < 2; ++k) {
        if ((rc = dalloc_e(c, &c->chainL[k], en * c->cap))) return rc;
        if ((rc = dalloc_e(c, &c->chainR[k], en * c->cap))) return rc;
        if (c->typed && ((rc = dalloc(c, &c->xchainL[k], en)) || (rc = dalloc(c, &c->xchainR[k], en)))) return rc;
    }
    if ((rc = dalloc(c, &c->yeval, en * c->C * (c->typed ? 2 : 1)))) return rc;      // typed: (re, im) pairs
    if ((rc = dalloc(c, &c->out3, 4))) return rc;
    if ((rc = dalloc(c, &c->conf, (int64_t)MAX_C * MAX_C))) return rc;
    if ((rc = dalloc(c, &c->pred, en))) return rc;
    c->eval_ready = true;
    return 0;
}

// the element-typed context's workspace (mpst_typed.hip): (E) buffers in the element type, everything that decides the
// truncation in fp64
int ensure_workspace_typed(Ctx* c) {
    const DataSet& tr = c->ds[MPST_TRAIN];
    const int dm = c->d * c->cap, zw = c->zw;
    const int64_t Lmax = (int64_t)dm * dm;
    int rc;
    if (zw * dm > DIM_LIMIT || zw * c->cap > CAP_LIMIT)
        return fail(c, MPST_ERR_UNSUPPORTED, "complex element type: 2*d*chi_max = %d (2*chi_max = %d) exceeds the eigensolver's limits %d, %d", zw * dm, zw * c->cap, DIM_LIMIT, CAP_LIMIT);
    if (c->d > 32) return fail(c, MPST_ERR_UNSUPPORTED, "the element-typed sweep holds d <= 32");
    TView tv = make_tview(c, MPST_TRAIN);
    if (typed_max_lds(tv) > 144 * 1024) return fail(c, MPST_ERR_UNSUPPORTED, "chi_max = %d, d = %d exceed the LDS staging of the element-typed kernels for this element type", c->cap, c->d);
    c->fused = false;
    c->b2 = false;
    if ((rc = dalloc_e(c, &c->LE, c->cache_elems))) return rc;
    if ((rc = dalloc_e(c, &c->RE, c->cache_elems))) return rc;
    if ((rc = dalloc(c, &c->xLE, (int64_t)c->T * tr.N)) || (rc = dalloc(c, &c->xRE, (int64_t)c->T * tr.N)) || (rc = dalloc(c, &c->yexp, tr.N))) return rc;
    if ((rc = dalloc_e(c, &c->bt, c->C * Lmax))) return rc;
    if ((rc = dalloc(c, &c->yhat, (int64_t)2 * c->C * tr.N))) return rc;
    if ((rc = dalloc(c, &c->tile_loss, std::max<int64_t>((int64_t)c->C * tr.ntiles, 1)))) return rc;
    c->partial_elems = (int64_t)c->C * typed_grad_nsplit(tv, tr.nchunks) * Lmax;
    if ((rc = dalloc_e(c, &c->partial, c->partial_elems))) return rc;
    c->n_norm_part = typed_norm_parts(tv);
    if ((rc = dalloc(c, &c->norm_part, c->n_norm_part))) return rc;
    if ((rc = dalloc(c, &c->loss_trace, (int64_t)2 * (c->T - 1) * (c->opt.update_iters + 1)))) return rc;
    HIPC(c, hipMemset(c->loss_trace, 0, (size_t)2 * (c->T - 1) * (c->opt.update_iters + 1) * sizeof(double)));
    if ((rc = dalloc(c, &c->gradbuf, 2 + c->C * Lmax * zw))) return rc;
    HIPC(c, hipMemset(c->gradbuf, 0, (size_t)(2 + c->C * Lmax * zw) * sizeof(double)));
    const int ne = std::max(zw * dm, MAX_DIM);
    if ((rc = dalloc(c, &c->gram, (int64_t)ne * ne))) return rc;
    if ((rc = dalloc(c, &c->lam, ne + 2))) return rc;
    if ((rc = dalloc(c, &c->E, (int64_t)ne * zw * c->cap))) return rc;
    if ((rc = dalloc(c, &c->tnorm_scratch, (int64_t)6 * c->cap * c->cap))) return rc;
    if (c->big) { big_eig_destroy(c->big); c->big = nullptr; }
    if (c->blk) { blocked_eig_destroy(c->blk); c->blk = nullptr; }
    dfree(&c->snap_sites); dfree(&c->snap_chi); dfree(&c->snap_sc);
    c->big_opt = false;
    if (zw * dm > MAX_DIM) {
        std::string e;
        if ((rc = big_eig_create(&c->big, zw * dm, c->stream, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        const char* sel = getenv("MPST_BIG_EIG");
        if (!(sel && (strcmp(sel, "jacobi") == 0 || strcmp(sel, "rocsolver") == 0)) && (rc = blocked_eig_create(&c->blk, zw * dm, &e)))
            return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        // real element types: the randomised subspace solver in front of the exact one (complex Gram matrices arrive as embeddings)
        if (c->blk && (rc = blocked_eig_enable_subspace(c->blk, zw * c->C * dm, c->cap, c->C, zw == 2, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        c->big_opt = c->blk && getenv("MPST_BIG_SYNC") == nullptr && getenv("MPST_BT_NO_COOP") == nullptr;
        if (const char* ff = getenv("MPST_BIG_FORCE_FAIL")) c->big_force_fail = atoi(ff);
    }
    if ((rc = dalloc(c, &c->eig_ws, (int64_t)eig_workspace_doubles()))) return rc;
    HIPC(c, hipMemset(c->eig_ws, 0, eig_workspace_doubles() * sizeof(double)));
    if ((rc = dalloc(c, &c->sc, 1))) return rc;
    HIPC(c, hipMemset(c->sc, 0, sizeof(DevScalars)));
    if ((rc = dalloc(c, &c->norm2, 1))) return rc;
    hipError_t ea = init_kernel_attrs(c->device);
    if (ea == hipSuccess) ea = eig_init_attrs(c->device);
    if (ea == hipSuccess) ea = typed_init_attrs(c->device);
    if (ea != hipSuccess) return fail(c, MPST_ERR_DEVICE, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));
    c->ws_ready = true;
    c->eval_ready = false;
    c->epoch++;
    return ensure_eval(c);
}

// (re)allocate everything whose size depends on (training set, options, capacity)
int ensure_workspace(Ctx* c) {
    if (!c->have_opt) return fail(c, MPST_ERR_INVALID, "mpst_set_options must be called first");
    if (!c->have_mps) return fail(c, MPST_ERR_INVALID, "mpst_set_mps must be called first");
    if (c->ws_ready) return ensure_eval(c);
    const DataSet& tr = c->ds[MPST_TRAIN];
    if (tr.N <= 0) return fail(c, MPST_ERR_INVALID, "no training data set (mpst_set_dataset)");
    const int dm = c->d * c->cap;
    if (dm > DIM_LIMIT || c->cap > CAP_LIMIT)
        return fail(c, MPST_ERR_UNSUPPORTED, "d*chi_max = %d (chi_max = %d) exceeds the engine's limits d*chi_max <= %d, chi_max <= %d", dm, c->cap, DIM_LIMIT, CAP_LIMIT);
    if (c->C > MAX_C) return fail(c, MPST_ERR_UNSUPPORTED, "more than %d classes unsupported", MAX_C);
    const int64_t Lmax = (int64_t)dm * dm;
    int rc;
    if (c->ipc_local && 2 + c->C * Lmax * c->zw > c->ipc_slot) ipc_release(c);   // inbox slots too small for the new capacity: export again
    c->caches_valid = false;
    c->cache_elems = (int64_t)c->T * tr.N * c->cap;
    if (c->typed) return ensure_workspace_typed(c);
    if ((rc = dalloc(c, &c->LE, c->cache_elems))) return rc;
    if ((rc = dalloc(c, &c->RE, c->cache_elems))) return rc;
    if ((rc = dalloc(c, &c->bt, c->C * Lmax))) return rc;
    if ((rc = dalloc(c, &c->yhat, (int64_t)c->C * tr.N))) return rc;
    if ((rc = dalloc(c, &c->tile_loss, std::max<int64_t>((int64_t)c->C * tr.ntiles, std::max(tr.nparts[0], tr.nparts[1])))) ) return rc;
    c->fused = dm <= MAX_DIM && !c->opt.rescale_before && getenv("MPST_NO_FUSED") == nullptr;
    const int pk = c->opt.loss == MPST_LOSS_MSE ? 1 : 0;
    (void)pk;
    // Which pair forms the gradient on the fused chain.  The sliced kernels (k_yhat_s + k_grad_s) move tens of KB per workgroup
    // and no partial gradients: 26 us against 31 us per bond at N = 4096, 11.6 MB against 40 MB of HBM traffic.  Their cost per
    // series is higher though (every series is read by 8 slice- and 16 block-workgroups: 2.7 against 1.5 us per 1000 series),
    // so from about 8000 series per rank the persistent k_bond_fused + k_fused_reduce pair wins (N = 32768: 75 against 105 us;
    // profiles/r03_*).  MPST_B2=0 / 1 forces either.
    {
        const char* e = getenv("MPST_B2");
        const bool want = e ? atoi(e) != 0 : (tr.N <= 8192 && getenv("MPST_NO_B2") == nullptr);
        c->b2 = c->fused && c->d >= 2 && c->d <= 16 && want;
    }
    if (c->fused) {
        c->partial_elems = (int64_t)std::max(tr.nparts[0], tr.nparts[1]) * Lmax;   // independent of N: one partial per persistent workgroup
        if (c->b2) {
            View gv{};
            gv.C = c->C; gv.d = c->d; gv.cap = c->cap;
            const int64_t max_pass = tr.N;                  // MSE walks every series in every pass; KLD at most that
            c->b2_ksplit = b2_ksplit(gv, max_pass);
            // a context that runs in batches of K fits shares the chip with K - 1 others: fewer, longer shares per gradient block
            // (less hand-over per fit; the share count fixes the order of the partial sums, so it belongs to the context, not to the call)
            // (d = 4: the batched launches run k_grad_s with four waves per workgroup, two workgroups per CU - twice the workgroups fill the chip)
            if (c->batch_hint > 1 && getenv("MPST_B2_KSPLIT") == nullptr)
                c->b2_ksplit = c->d == 4 ? std::max(1, std::min(c->b2_ksplit, 2 * c->b2_ksplit / std::min(c->batch_hint, 16)))
                                         : std::max(1, c->b2_ksplit / std::min(c->batch_hint, 8));
            c->b2_norm_parts = c->C * b2_blocks_cap(gv);
            c->partial_elems = std::max(c->partial_elems, b2_partial_elems(gv, max_pass));
            if ((rc = dalloc(c, &c->b2_ypart, (int64_t)8 * c->C * tr.N))) return rc;
            if ((rc = dalloc(c, &c->b2_lossp, (int64_t)c->C * 64))) return rc;     // GS_MAXKS shares per class
            if ((rc = dalloc(c, &c->b2_tick, (int64_t)c->b2_norm_parts + 1))) return rc;
            HIPC(c, hipMemset(c->b2_tick, 0, (size_t)(c->b2_norm_parts + 1) * sizeof(unsigned int)));
#ifdef MPST_B2_DEBUG
            if ((rc = dalloc(c, &c->b2_dbg, 8192 * 8))) return rc;
            HIPC(c, hipMemset(c->b2_dbg, 0, 8192 * 8 * sizeof(unsigned long long)));
#endif
        }
    } else {
        const int nbcap = ((dm + GB - 1) / GB) * ((dm + GB - 1) / GB);
        c->partial_elems = (int64_t)c->C * grad_nsplit(tr.nchunks, nbcap, c->C) * Lmax;    // one partial per k_grad workgroup share, independent of N
    }
    if ((rc = dalloc(c, &c->partial, c->partial_elems))) return rc;
    if ((rc = dalloc(c, &c->btn, c->C * Lmax))) return rc;
    {
        const char* e4 = getenv("MPST_CHAIN4");
        // (a context that is advanced in batches keeps the six-launch chain mpst_sweep_batch runs: its solo and its batched sweeps agree bit for bit)
        c->chain4_ok = c->fused && c->b2 && c->batch_hint <= 1 && !(e4 && e4[0] == '0');
        c->ynext_lid = -1;
        c->tail_launches = 0;
        c->tail_force_fail = -1;
        if (const char* ff = getenv("MPST_TAIL_FORCE_REDO")) c->tail_force_fail = atoi(ff);
        dfree(&c->btnT);
        if (c->chain4_ok && (rc = dalloc(c, &c->btnT, c->C * Lmax))) return rc;
        if (c->chain4_ok && (rc = dalloc(c, &c->tail_span, 2 * 2048))) return rc;
        if (c->chain4_ok) HIPC(c, hipMemset(c->tail_span, 0, 2 * 2048 * sizeof(unsigned long long)));
    }
    c->n_norm_part = (int)((c->C * Lmax + 63) / 64);      // RED_E entries per workgroup of k_fused_reduce
    if ((rc = dalloc(c, &c->loss_trace, (int64_t)2 * (c->T - 1) * (c->opt.update_iters + 1)))) return rc;
    HIPC(c, hipMemset(c->loss_trace, 0, (size_t)2 * (c->T - 1) * (c->opt.update_iters + 1) * sizeof(double)));
    if ((rc = dalloc(c, &c->norm_part, std::max(c->n_norm_part, c->b2_norm_parts)))) return rc;
    if ((rc = dalloc(c, &c->gradbuf, 2 + c->C * Lmax))) return rc;
    HIPC(c, hipMemset(c->gradbuf, 0, (size_t)(2 + c->C * Lmax) * sizeof(double)));
    const int dmx = std::max(dm, MAX_DIM);
    if ((rc = dalloc(c, &c->gram, (int64_t)dmx * dmx))) return rc;
    if ((rc = dalloc(c, &c->lam, dmx + 2))) return rc;
    if ((rc = dalloc(c, &c->E, (int64_t)dmx * c->cap))) return rc;
    if ((rc = dalloc(c, &c->norm_scratch, (int64_t)3 * c->cap * c->cap))) return rc;
    if (c->big) { big_eig_destroy(c->big); c->big = nullptr; }
    if (c->blk) { blocked_eig_destroy(c->blk); c->blk = nullptr; }
    dfree(&c->snap_sites); dfree(&c->snap_chi); dfree(&c->snap_sc);       // sized for the previous MPS
    c->big_opt = false;
    if (dm > MAX_DIM) {
        std::string e;
        if ((rc = big_eig_create(&c->big, dm, c->stream, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        const char* sel = getenv("MPST_BIG_EIG");
        if (!(sel && (strcmp(sel, "jacobi") == 0 || strcmp(sel, "rocsolver") == 0)) && (rc = blocked_eig_create(&c->blk, dm, &e)))
            return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        // MPST_BIG_SYNC=1: read the eigensolver's verdict after every bond (one host synchronisation per bond) instead of once per sweep
        if (c->blk && (rc = blocked_eig_enable_subspace(c->blk, c->C * dm, c->cap, c->C, 0, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
        c->big_opt = c->blk && getenv("MPST_BIG_SYNC") == nullptr && getenv("MPST_BT_NO_COOP") == nullptr;
        if (const char* ff = getenv("MPST_BIG_FORCE_FAIL")) c->big_force_fail = atoi(ff);       // test hook: the n-th solve of the context is marked failed
    }
    if ((rc = dalloc(c, &c->eig_ws, (int64_t)eig_workspace_doubles()))) return rc;
    HIPC(c, hipMemset(c->eig_ws, 0, eig_workspace_doubles() * sizeof(double)));
    if ((rc = dalloc(c, &c->sc, 1))) return rc;
    HIPC(c, hipMemset(c->sc, 0, sizeof(DevScalars)));
    if ((rc = dalloc(c, &c->norm2, 1))) return rc;
    hipError_t ea = init_kernel_attrs(c->device);
    if (ea == hipSuccess) ea = eig_init_attrs(c->device);
    if (ea == hipSuccess) ea = b2_init_attrs(c->device);
    if (ea != hipSuccess) return fail(c, MPST_ERR_DEVICE, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));
    c->ws_ready = true;
    c->eval_ready = false;
    c->epoch++;
    return ensure_eval(c);
}

// clears the sticky error flag and the per-sweep diagnostics (status, eig_sweeps_total, eig_fallbacks are adjacent);
// enqueued at the head of every sweep / bond step - inside the captured graph too - so that a context recovers
// after a failed decomposition (the class of failure tune() retries on)
// The subspace eigensolver's counters, read where the stream has just been synchronised (after a sweep, a bond step, a batch): the info
// query returns these and never touches the stream.
static void refresh_ss_counts(Ctx* c) {
    if (c->blk) (void)blocked_eig_subspace_counts(c->blk, c->stream, &c->ss_counts[0], &c->ss_counts[1]);
}
int enqueue_reset_status(Ctx* c) {
    static_assert(offsetof(DevScalars, eig_fallbacks) == offsetof(DevScalars, status) + 8, "status / eig_sweeps_total / eig_fallbacks adjacent");
    HIPC(c, hipMemsetAsync((char*)c->sc + offsetof(DevScalars, status), 0, 12, c->stream));
    HIPC(c, hipMemsetAsync((char*)c->sc + offsetof(DevScalars, redo), 0, 4, c->stream));
    return 0;
}

// ---- profiling helpers: HIP events on the stream the kernels are launched on -------------
hipEvent_t pool_event(Ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}
struct ProfScope {
    Ctx* c; int k; bool on; size_t e0 = 0;
    ProfScope(Ctx* c_, int k_) : c(c_), k(k_), on((c_->prof_mask >> k_) & 1u) {
        if (on) { e0 = c->ev_used; (void)hipEventRecord(pool_event(c), c->stream); }
    }
    ~ProfScope() {
        if (on) { size_t e1 = c->ev_used; (void)hipEventRecord(pool_event(c), c->stream); c->recs.push_back({k, e0, e1}); }
    }
};
void prof_collect(Ctx* c) {
    for (auto& r : c->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e1]) == hipSuccess) {
            c->prof_us[r.k] += 1e3 * ms;
            c->prof_cnt[r.k] += 1;
        }
    }
    c->recs.clear();
    c->ev_used = 0;
}

// ---- sum over ranks of a device buffer: the one-shot path when the inboxes are attached, else RCCL ---------------
void ipc_release(struct Ctx* c) {
    for (int r = 0; r < AR_MAX_RANKS; ++r) {
        if (c->ipc_peer[r] && c->ipc_peer[r] != c->ipc_local) (void)hipIpcCloseMemHandle(c->ipc_peer[r]);
        c->ipc_peer[r] = nullptr;
    }
    if (c->ipc_local) (void)hipFree(c->ipc_local);
    c->ipc_local = nullptr;
    c->use_ipc = false;
}
int enqueue_allreduce(Ctx* c, double* buf, int64_t n_fixed, int lid) {
    if (!multi(c)) return 0;
    if (c->use_ipc) {
        ArParams p{};
        for (int r = 0; r < c->nranks; ++r) {
            p.inbox[r] = (double*)c->ipc_peer[r];
            p.flags[r] = (unsigned long long*)((char*)c->ipc_peer[r] + c->ipc_flag_off);
        }
        p.buf = buf;
        p.status = &c->sc->status;
        p.dbg = c->sc->pad;
        p.counter = (unsigned int*)((char*)c->ipc_local + c->ipc_ctr_off);
        p.nranks = c->nranks; p.rank = c->rank; p.slot = c->ipc_slot;
        p.epoch = ++c->ar_epoch;
        {
            static const double secs = [] { const char* e = getenv("MPST_AR_TIMEOUT_S"); const double v = e ? atof(e) : 10.0; return v > 0.0 ? v : 10.0; }();
            p.spin_limit = (long long)(secs * 2.4e9);
        }
        p.chi = c->chi; p.lid = lid; p.C = c->C * c->zw; p.d = c->d; p.n_fixed = n_fixed;      // complex gradients: (re, im) pairs
        if (lid < 0 && n_fixed > c->ipc_slot) return fail(c, MPST_ERR_INVALID, "all-reduce message exceeds the inbox slot");
        launch_allreduce_oneshot(p, c->stream);
        return 0;
    }
    if (c->ipc_dead && !c->comm)
        return fail(c, MPST_ERR_DEVICE, "the one-shot all-reduce timed out earlier: export and attach the inboxes again (mpst_comm_ipc_export / _attach)");
    if (!c->comm) return fail(c, MPST_ERR_INVALID, "%d ranks but neither an RCCL communicator nor attached inboxes", c->nranks);
    const size_t cnt = lid >= 0 ? 2 + (size_t)c->zw * c->C * c->d * c->cap * c->d * c->cap : (size_t)n_fixed;
    Rccl* nc = rccl_ready(nullptr);
    if (!nc) return fail(c, MPST_ERR_DEVICE, "RCCL is not available");
    ncclResult_t r = nc->AllReduce(buf, buf, cnt, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail(c, MPST_ERR_DEVICE, "ncclAllReduce: %s", nc->GetErrorString(r));
    return 0;
}

// ---- eigensolver of a bond whose Gram matrix exceeds the LDS-resident solver (both element paths) -----------------------
int enqueue_big_eig(Ctx* c, const View& v, int lid, int going_left) {
    hipStream_t s = c->stream;
    int need_lib = 1;
    if (c->blk && c->big_opt_active) {
        if (launch_eig_blocked_nosync(v, lid, going_left, c->blk, s)) return fail(c, MPST_ERR_DEVICE, "blocked eigensolver failed at bond %d: %s", lid, hipGetErrorString(hipGetLastError()));
        if (c->big_force_fail >= 0 && c->big_solves++ == c->big_force_fail) blocked_eig_force_sticky(c->blk, s);      // test hook
        need_lib = 0;
    } else if (c->blk) {
        need_lib = launch_eig_blocked(v, lid, going_left, nullptr, 0, nullptr, nullptr, nullptr, c->blk, s);
        if (need_lib < 0) return fail(c, MPST_ERR_DEVICE, "blocked eigensolver failed at bond %d: %s", lid, hipGetErrorString(hipGetLastError()));
        if (need_lib) c->big_fallbacks++;
    }
    if (need_lib && launch_eig_big(v, lid, going_left, c->big, s)) return fail(c, MPST_ERR_DEVICE, "the large-bond Jacobi fallback could not be launched at bond %d", lid);
    return 0;
}

// ---- the per-bond chain of the element-typed sweep (mpst_typed.hip) -----------------------------------------------------
int enqueue_bond_typed(Ctx* c, int lid, int going_left, int trace_row) {
    hipStream_t s = c->stream;
    TView t = make_tview(c, MPST_TRAIN);
    const int n_it = c->opt.update_iters, rid = lid + 1;
    if (c->opt.track_cost && trace_row >= 0) t.trace = c->loss_trace + (int64_t)trace_row * (n_it + 1);
    { ProfScope p(c, K_BT); launch_tbt_assemble(t, lid, s); }                    // flatten_bt
    if (t.rescale_before) launch_tbt_prescale(t, lid, s);
    for (int it = 0; it < n_it; ++it) {
        { ProfScope p(c, K_YHAT); launch_tyhat(t, lid, s); }
        { ProfScope p(c, K_GRAD); launch_tgrad(t, lid, s); }
        { ProfScope p(c, K_UPDATE); launch_tgrad_reduce(t, lid, s); }
        if (multi(c)) {
            ProfScope p(c, K_ALLREDUCE);
            int rc = enqueue_allreduce(c, c->gradbuf, 0, lid);
            if (rc) return rc;
            launch_tgrad_norm(t, lid, s);
        }
        t.trace_it = it;
        { ProfScope p(c, K_UPDATE); launch_tupdate(t, lid, it == 0, s); }
    }
    { ProfScope p(c, K_GRAM); launch_tgram(t, lid, going_left, s); }             // decomposeBT: Gram matrix, fp64
    const View ve = make_eig_view(c);
    if (c->big) {
        ProfScope p(c, K_EIG_TRI);
        int rc = enqueue_big_eig(c, ve, lid, going_left);
        if (rc) return rc;
    } else {
        { ProfScope p(c, K_EIG_TRI); launch_eig(ve, lid, going_left, 0, s); }
        if (!eig_merged()) { ProfScope p(c, K_EIG_VEC); launch_eig(ve, lid, going_left, 1, s); }
        { ProfScope p(c, K_EIG_FIN); launch_eig(ve, lid, going_left, 2, s); }
    }
    if (t.trace) {          // track_cost: the loss at the updated (and, with rescale[2], normalised) bond tensor
        TView ty = t;
        ty.yhat_scaled = t.rescale_after;
        launch_tyhat(ty, lid, s);
        View vt = make_view(c, MPST_TRAIN);
        vt.trace = t.trace;
        vt.trace_it = n_it;
        launch_trace_loss(vt, s);
        if (multi(c)) {
            int rc = enqueue_allreduce(c, vt.trace + n_it, 1, -1);
            if (rc) return rc;
        }
    }
    { ProfScope p(c, K_SPLIT); launch_tsplit(t, lid, going_left, s); }
    {
        ProfScope p(c, K_ENV);                                                   // update_caches!: the new site tensor is the map
        const int64_t cs = (int64_t)t.N * t.cap;
        if (going_left)
            launch_tenv(t, rid, 0, rid < c->T - 1 ? eoff(c, c->RE, (int64_t)(rid + 1) * cs) : nullptr, rid < c->T - 1 ? c->xRE + (int64_t)(rid + 1) * t.N : nullptr, rid + 1,
                        ENV_M_SITE_T, rid, eoff(c, c->RE, (int64_t)rid * cs), c->xRE + (int64_t)rid * t.N, s);
        else
            launch_tenv(t, lid, 1, lid > 0 ? eoff(c, c->LE, (int64_t)(lid - 1) * cs) : nullptr, lid > 0 ? c->xLE + (int64_t)(lid - 1) * t.N : nullptr, lid, ENV_M_SITE, lid + 1,
                        eoff(c, c->LE, (int64_t)lid * cs), c->xLE + (int64_t)lid * t.N, s);
    }
    return 0;
}
// construct_caches of the typed context: left environments of sites [0, upto), right environments of sites (from, T-1]
void enqueue_caches_typed(Ctx* c, int left_upto, int right_from) {
    TView t = make_tview(c, MPST_TRAIN);
    const int64_t cs = (int64_t)t.N * t.cap;
    ProfScope p(c, K_ENV);
    for (int j = 0; j < left_upto && j <= c->T - 2; ++j)
        launch_tenv(t, j, 1, j > 0 ? eoff(c, c->LE, (int64_t)(j - 1) * cs) : nullptr, j > 0 ? c->xLE + (int64_t)(j - 1) * t.N : nullptr, j, ENV_M_SITE, j + 1,
                    eoff(c, c->LE, (int64_t)j * cs), c->xLE + (int64_t)j * t.N, c->stream);
    for (int j = c->T - 1; j > right_from && j >= 1; --j)
        launch_tenv(t, j, 0, j < c->T - 1 ? eoff(c, c->RE, (int64_t)(j + 1) * cs) : nullptr, j < c->T - 1 ? c->xRE + (int64_t)(j + 1) * t.N : nullptr, j + 1, ENV_M_SITE_T, j,
                    eoff(c, c->RE, (int64_t)j * cs), c->xRE + (int64_t)j * t.N, c->stream);
}

// ---- the per-bond launch chain (RealRealHighDimension.jl:733-762 / :777-801) ---------------
// have_bt: the bond tensor of this bond was already assembled by the previous bond's environment
// kernel; next_bt_lid >= 0: assemble that bond's tensor inside this bond's environment kernel.
int enqueue_bond(Ctx* c, const View& v_in, int lid, int going_left, bool have_bt = false, int next_bt_lid = -1, int trace_row = -1) {
    if (c->typed) return enqueue_bond_typed(c, lid, going_left, trace_row);
    hipStream_t s = c->stream;
    View v = v_in;
    const int n_it = c->opt.update_iters;
    if (c->opt.track_cost && trace_row >= 0) v.trace = c->loss_trace + (int64_t)trace_row * (n_it + 1);
    // track_cost: the loss at the updated (and, with rescale[2], normalised) bond tensor - one more forward pass over the batch
    auto trace_final = [&](const double* bt_new) -> int {
        if (!v.trace) return 0;
        View vy = v;
        vy.bt = const_cast<double*>(bt_new);
        vy.yhat_scaled = v.rescale_after;
        vy.trace_it = n_it;
        launch_yhat(vy, lid, s);
        launch_trace_loss(vy, s);
        // a shard's tile losses carry the GLOBAL 1/N: the sum over the ranks is the loss the single-rank run records
        if (multi(c)) return enqueue_allreduce(c, vy.trace + n_it, 1, -1);
        return 0;
    };
    const int rid = lid + 1;
    const uint64_t seq_before = c->bond_seq++;
    if (c->fused) {
        const int iters = c->opt.update_iters;
        // the next bond's tensor is assembled by this bond's last launch when the sweep moves on in the same direction
        const int chain = (next_bt_lid >= 0 && next_bt_lid == (going_left ? lid - 1 : lid + 1)) ? 1 : 0;
        // four launches per bond: k_eig_fin, k_env_split and the NEXT bond's k_yhat_s in one (k_bond_tail, mpst_fused.hip)
        const bool use4 = c->chain4_ok && !c->chain4_hold && c->b2 && !multi(c) && iters == 1 && !v.trace && eig_merged() && bond_tail_supported(v);
        // ... whose overlaps are this bond's if the launch before this one was the neighbouring bond's tail
        const bool y_ready = use4 && c->ynext_lid == lid && c->ynext_epoch == c->epoch && c->ynext_seq == seq_before;
        c->ynext_lid = -1;
        if (!have_bt) { ProfScope p(c, K_BT); launch_bt_assemble(v, lid, s); }   // flatten_bt :733/:777
        View vl = v;                      // after k_grad_s on one rank the loss is still in pieces (bond_loss)
        vl.n_lossp = c->b2 ? c->b2_ksplit : 0;
        for (int it = 0; it < iters; ++it) {                                     // TSGO/custGD :44,:75
            if (c->b2) {
                if (!(y_ready && it == 0)) { ProfScope p(c, K_YHAT); launch_yhat_s(v, lid, s); }           // yhat, by column slices of B_c
                { ProfScope p(c, K_GRAD); launch_grad_s(v, lid, s); }           // gradient blocks, reduced by their last arriver
                if (multi(c)) launch_loss_sum(vl, s);                      // the loss travels in gradbuf[0]
            } else {
                { ProfScope p(c, K_GRAD); launch_bond_fused(v, lid, 0, s); }    // yhat + gradient partials
                { ProfScope p(c, K_UPDATE); launch_fused_reduce(v, lid, s); }
            }
            if (multi(c)) {
                ProfScope p(c, K_ALLREDUCE);
                int rc = enqueue_allreduce(c, c->gradbuf, 0, lid);
                if (rc) return rc;
                launch_grad_norm(v, lid, s);
            }
            v.trace_it = it;
            if (it + 1 < iters) {
                ProfScope p(c, K_UPDATE);
                View vu = v;
                if (c->b2 && !multi(c)) vu.n_lossp = c->b2_ksplit;
                launch_update(vu, lid, it == 0, s);
            }
        }
        {
            ProfScope p(c, K_GRAM);                                               // last step + decomposeBT :756/:798
            View vg = v;
            if (c->b2 && !multi(c)) vg.n_lossp = c->b2_ksplit;
            // pieces of ||grad||^2: one per gradient block from k_grad_s; after an all-reduce k_grad_norm has rewritten them
            if (c->b2 && !multi(c)) vg.n_norm_part = c->b2_norm_parts;
            if (!use4) vg.btnT = nullptr;
            launch_gram_upd(vg, lid, going_left, iters == 1, s);
        }
        { ProfScope p(c, K_EIG_TRI); launch_eig(v, lid, going_left, 0, s); }
        if (use4) {
            ProfScope p(c, K_ENV);                                                // verification, back-split, update_caches!, next yhat
            const int nxt = going_left ? lid - 1 : lid + 1;
            const int want = (nxt >= 0 && nxt <= c->T - 2) ? 1 : 0;
            launch_bond_tail(v, lid, going_left, chain, want | (c->tail_launches++ == c->tail_force_fail ? 4 : 0), c->tail_span, s);
            if (want) {
                c->ynext_lid = nxt;
                c->ynext_epoch = c->epoch;
                c->ynext_seq = c->bond_seq;
            }
            return 0;
        }
        if (!eig_merged()) { ProfScope p(c, K_EIG_VEC); launch_eig(v, lid, going_left, 1, s); }
        { ProfScope p(c, K_EIG_FIN); launch_eig(v, lid, going_left, 2, s); }
        { int rc = trace_final(c->btn); if (rc) return rc; }
        {
            ProfScope p(c, K_ENV);                                                // back-split + update_caches! :759/:799
            const int64_t cs = (int64_t)v.N * v.cap;
            if (going_left)
                launch_env_split(v, lid, 1, rid, 0, rid < c->T - 1 ? c->RE + (int64_t)(rid + 1) * cs : nullptr, rid + 1, rid,
                                 c->RE + (int64_t)rid * cs, chain, s);
            else
                launch_env_split(v, lid, 0, lid, 1, lid > 0 ? c->LE + (int64_t)(lid - 1) * cs : nullptr, lid, lid + 1,
                                 c->LE + (int64_t)lid * cs, chain, s);
        }
        return 0;
    }
    if (!have_bt) { ProfScope p(c, K_BT); launch_bt_assemble(v, lid, s); }      // flatten_bt :733/:777
    if (v.rescale_before) launch_bt_prescale(v, lid, s);                        // loss_functions.jl:109
    for (int it = 0; it < c->opt.update_iters; ++it) {                          // TSGO/custGD :44,:75
        { ProfScope p(c, K_YHAT); launch_yhat(v, lid, s); }
        { ProfScope p(c, K_GRAD); launch_grad(v, lid, s); }
        { ProfScope p(c, K_UPDATE); launch_grad_reduce(v, lid, s); }
        if (multi(c)) {
            ProfScope p(c, K_ALLREDUCE);
            int rc = enqueue_allreduce(c, c->gradbuf, 0, lid);
            if (rc) return rc;
        }
        v.trace_it = it;
        { ProfScope p(c, K_UPDATE); launch_update(v, lid, it == 0, s); }
    }
    { ProfScope p(c, K_GRAM); launch_gram(v, lid, going_left, s); }            // decomposeBT :756/:798
    if (c->big) {
        ProfScope p(c, K_EIG_TRI);
        int rc = enqueue_big_eig(c, v, lid, going_left);
        if (rc) return rc;
    } else {
        { ProfScope p(c, K_EIG_TRI); launch_eig(v, lid, going_left, 0, s); }
        if (!eig_merged()) { ProfScope p(c, K_EIG_VEC); launch_eig(v, lid, going_left, 1, s); }
        { ProfScope p(c, K_EIG_FIN); launch_eig(v, lid, going_left, 2, s); }
    }
    { int rc = trace_final(c->bt); if (rc) return rc; }
    { ProfScope p(c, K_SPLIT); launch_split(v, lid, going_left, s); }
    {
        ProfScope p(c, K_ENV);                                                  // update_caches! :759/:799
        const int64_t cs = (int64_t)v.N * v.cap;
        if (going_left)
            launch_env(v, rid, 0, rid < c->T - 1 ? c->RE + (int64_t)(rid + 1) * cs : nullptr, rid + 1, ENV_M_E, rid,
                       c->RE + (int64_t)rid * cs, s, next_bt_lid);
        else
            launch_env(v, lid, 1, lid > 0 ? c->LE + (int64_t)(lid - 1) * cs : nullptr, lid, ENV_M_E, lid + 1,
                       c->LE + (int64_t)lid * cs, s, next_bt_lid);
    }
    return 0;
}

// MPST_ENV_WALK=0: construct_caches as one k_env launch per site (the same bits)
bool env_walk_on(const Ctx* c, const View& v) {
    const char* e = getenv("MPST_ENV_WALK");
    return !c->typed && !(e && e[0] == '0') && env_walk_supported(v);
}

// construct_caches (RealRealHighDimension.jl:45-103)
void enqueue_caches(Ctx* c, const View& v, int going_left) {
    if (c->typed) {
        if (going_left) enqueue_caches_typed(c, c->T - 1, c->T);
        else enqueue_caches_typed(c, 0, 0);
        return;
    }
    const int64_t cs = (int64_t)v.N * v.cap;
    ProfScope p(c, K_ENV);
    if (env_walk_on(c, v)) {           // one launch: every workgroup walks the chain with its 16 series (k_env_walk, the bits of the per-site launches)
        launch_env_walk(v, going_left, c->T - 1, c->stream);
        return;
    }
    if (going_left) {
        for (int j = 0; j <= c->T - 2; ++j)
            launch_env(v, j, 1, j > 0 ? c->LE + (int64_t)(j - 1) * cs : nullptr, j, ENV_M_SITE, j + 1,
                       c->LE + (int64_t)j * cs, c->stream);
    } else {
        for (int j = c->T - 1; j >= 1; --j)
            launch_env(v, j, 0, j < c->T - 1 ? c->RE + (int64_t)(j + 1) * cs : nullptr, j + 1, ENV_M_SITE_T, j,
                       c->RE + (int64_t)j * cs, c->stream);
    }
}

int check_ready(Ctx* c) {
    if (!c) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    return ensure_workspace(c);
}

int host_chi(Ctx* c, std::vector<int32_t>& chi, int32_t* ls) {
    chi.resize(c->T + 1);
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemcpy(chi.data(), c->chi, (size_t)(c->T + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPC(c, hipMemcpy(ls, c->label_site, sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

// evaluation chain into scratch: returns yhat on device in c->yeval ([N][C])
int enqueue_eval(Ctx* c, int which) {
    View v = make_view(c, which);
    if (v.N <= 0) return fail(c, MPST_ERR_INVALID, "data set %d is empty", which);
    std::vector<int32_t> chi; int32_t p;
    int rc = host_chi(c, chi, &p);
    if (rc) return rc;
    if (c->typed) {
        TView t = make_tview(c, which);
        const void* Lt = nullptr; const void* Rt = nullptr;
        const int32_t *Lx = nullptr, *Rx = nullptr;
        int q = 0;
        for (int j = 0; j < p; ++j) {
            launch_tenv(t, j, 1, j > 0 ? c->chainL[q ^ 1] : nullptr, j > 0 ? c->xchainL[q ^ 1] : nullptr, j, ENV_M_SITE, j + 1, c->chainL[q], c->xchainL[q], c->stream);
            Lt = c->chainL[q]; Lx = c->xchainL[q]; q ^= 1;
        }
        q = 0;
        for (int j = c->T - 1; j > p; --j) {
            launch_tenv(t, j, 0, j < c->T - 1 ? c->chainR[q ^ 1] : nullptr, j < c->T - 1 ? c->xchainR[q ^ 1] : nullptr, j + 1, ENV_M_SITE_T, j, c->chainR[q], c->xchainR[q], c->stream);
            Rt = c->chainR[q]; Rx = c->xchainR[q]; q ^= 1;
        }
        launch_teval_final(t, Lt, Lx, Rt, Rx, c->yeval, c->stream);
        return 0;
    }
    const double* Lc = nullptr; const double* Rc = nullptr;
    int pp = 0;
    for (int j = 0; j < p; ++j) {
        launch_env(v, j, 1, j > 0 ? c->chainL[pp ^ 1] : nullptr, j, ENV_M_SITE, j + 1, c->chainL[pp], c->stream);
        Lc = c->chainL[pp]; pp ^= 1;
    }
    pp = 0;
    for (int j = c->T - 1; j > p; --j) {
        launch_env(v, j, 0, j < c->T - 1 ? c->chainR[pp ^ 1] : nullptr, j + 1, ENV_M_SITE_T, j, c->chainR[pp], c->stream);
        Rc = c->chainR[pp]; pp ^= 1;
    }
    launch_eval_final(v, Lc, Rc, c->yeval, c->stream);
    return 0;
}

}  // namespace

// ===========================================================================================
extern "C" {

int mpst_version(void) { return MPST_ABI_VERSION; }

const char* mpst_last_error(void* ctx) { return ctx ? ((Ctx*)ctx)->err.c_str() : g_err.c_str(); }

int mpst_create(void** ctx, int device_id) {
    if (!ctx) return fail(nullptr, MPST_ERR_INVALID, "ctx is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, MPST_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, MPST_ERR_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    Ctx* c = new Ctx();
    static std::atomic<uint64_t> next_uid{1};
    c->uid = next_uid.fetch_add(1);
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return fail(nullptr, MPST_ERR_DEVICE, "cannot initialise device %d", device_id);
    }
    (void)hipEventCreate(&c->ev_start);
    (void)hipEventCreate(&c->ev_stop);
    *ctx = c;
    return 0;
}

void mpst_destroy(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->sweep_graph) (void)hipGraphExecDestroy(c->sweep_graph);
    if (c->batch_graph) (void)hipGraphExecDestroy(c->batch_graph);
    dfree(&c->batch_views);
    if (c->comm && rccl_ready(nullptr)) rccl_ready(nullptr)->CommDestroy(c->comm);
    ipc_release(c);
    free_dataset(c->ds[0]); free_dataset(c->ds[1]);
    dfree(&c->sites); dfree(&c->chi); dfree(&c->label_site); dfree(&c->LE); dfree(&c->RE); dfree(&c->bt);
    dfree(&c->yhat); dfree(&c->tile_loss); dfree(&c->partial); dfree(&c->gradbuf); dfree(&c->gram); dfree(&c->lam);
    if (c->big) big_eig_destroy(c->big);
    if (c->blk) blocked_eig_destroy(c->blk);
    dfree(&c->snap_sites); dfree(&c->snap_chi); dfree(&c->snap_sc);
    dfree(&c->norm_scratch); dfree(&c->tnorm_scratch); dfree(&c->xLE); dfree(&c->xRE); dfree(&c->yexp);
    for (int k = 0; k < 2; ++k) { dfree(&c->xchainL[k]); dfree(&c->xchainR[k]); }
    dfree(&c->btn); dfree(&c->norm_part); dfree(&c->loss_trace);
    dfree(&c->b2_ypart); dfree(&c->b2_lossp); dfree(&c->b2_tick); dfree(&c->b2_dbg);
    dfree(&c->btnT); dfree(&c->tail_span);
    dfree(&c->E); dfree(&c->eig_ws); dfree(&c->sc); dfree(&c->norm2); dfree(&c->yeval); dfree(&c->out3); dfree(&c->conf); dfree(&c->pred);
    for (int k = 0; k < 2; ++k) { dfree(&c->chainL[k]); dfree(&c->chainR[k]); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int mpst_comm_unique_id(uint8_t out_id[128]) {
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId size");
    std::string why;
    Rccl* nc = rccl_ready(&why);
    if (!nc) return fail(nullptr, MPST_ERR_DEVICE, "%s", why.c_str());
    ncclResult_t r = nc->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MPST_ERR_DEVICE, "ncclGetUniqueId: %s", nc->GetErrorString(r));
    memcpy(out_id, &id, 128);
    return 0;
}

int mpst_comm_init(void* ctx, const uint8_t unique_id[128], int nranks, int rank) {
    Ctx* c = (Ctx*)ctx;
    if (!c || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MPST_ERR_INVALID, "bad communicator arguments");
    HIPC(c, hipSetDevice(c->device));
    std::string why;
    Rccl* nc = rccl_ready(&why);
    if (!nc) return fail(c, MPST_ERR_DEVICE, "%s", why.c_str());
    // Only six entry points are used (ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / GetErrorString / GetVersion)
    // with ncclDouble, ncclSum and the 128-byte ncclUniqueId - unchanged across the 2.x series - so a host's copy of another
    // MINOR version (torch 2.10 ships 2.26.6 next to ROCm 7.2's 2.27.7) is accepted; another major version is not.
    if (nc->version / 10000 != NCCL_VERSION_CODE / 10000 || nc->version < 21800)
        return fail(c, MPST_ERR_DEVICE, "librccl %s has version %d, this library was built against %d (set MPST_RCCL_LIB to a matching one)",
                    nc->path.c_str(), nc->version, (int)NCCL_VERSION_CODE);
    if (c->comm) { nc->CommDestroy(c->comm); c->comm = nullptr; }
    c->nranks = nranks; c->rank = rank;
    c->force_coll = getenv("MPST_FORCE_COLLECTIVE") != nullptr;
    c->epoch++;
    ncclUniqueId id;
    memcpy(&id, unique_id, 128);
    ncclResult_t r = nc->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return fail(c, MPST_ERR_DEVICE, "ncclCommInitRank: %s", nc->GetErrorString(r)); }
    return 0;
}

int mpst_comm_library(char* path_out, int32_t path_cap, int32_t* version_out, int32_t* built_against_out) {
    Rccl* nc = rccl_get();
    if (version_out) *version_out = nc->version;
    if (built_against_out) *built_against_out = (int32_t)NCCL_VERSION_CODE;
    if (path_out && path_cap > 0) {
        const std::string t = nc->h ? nc->path + " (" + nc->how + ")" : nc->err;
        snprintf(path_out, (size_t)path_cap, "%s", t.c_str());
    }
    return nc->h ? 0 : MPST_ERR_DEVICE;
}

int mpst_comm_ipc_export(void* ctx, int nranks, int rank, uint8_t handle_out[64]) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !handle_out || nranks < 2 || nranks > AR_MAX_RANKS || rank < 0 || rank >= nranks)
        return fail(c, MPST_ERR_INVALID, "one-shot all-reduce: 2..%d ranks", AR_MAX_RANKS);
    int rc = check_ready(c);             // the slot size follows the gradient buffer: options, data set and MPS first
    if (rc) return rc;
    if (c->comm && (c->nranks != nranks || c->rank != rank)) return fail(c, MPST_ERR_INVALID, "rank / size disagree with the RCCL communicator");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    ipc_release(c);
    c->nranks = nranks; c->rank = rank;
    const int64_t dm = (int64_t)c->d * c->cap;
    c->ipc_slot = std::max<int64_t>(2 + c->zw * c->C * dm * dm, 4 + (int64_t)c->C * c->C);
    c->ipc_slot = (c->ipc_slot + 15) & ~15ll;
    c->ipc_flag_off = (size_t)2 * nranks * c->ipc_slot * sizeof(double);
    c->ipc_ctr_off = c->ipc_flag_off + 16 * sizeof(unsigned long long);
    c->ipc_bytes = c->ipc_ctr_off + 64;
    // fine-grained device memory: peers write it and this device polls it while kernels run
    hipError_t e = hipExtMallocWithFlags(&c->ipc_local, c->ipc_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail(c, MPST_ERR_NOMEM, "hipExtMallocWithFlags(fine-grained, %zu bytes): %s", c->ipc_bytes, hipGetErrorString(e));
    HIPC(c, hipMemset(c->ipc_local, 0, c->ipc_bytes));
    HIPC(c, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, c->ipc_local);
    if (e != hipSuccess) return fail(c, MPST_ERR_DEVICE, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle_out, &h, 64);
    c->ar_epoch = 0;
    c->epoch++;
    return 0;
}

int mpst_comm_ipc_attach(void* ctx, const uint8_t* all_handles) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !all_handles || !c->ipc_local) return fail(c, MPST_ERR_INVALID, "call mpst_comm_ipc_export first");
    HIPC(c, hipSetDevice(c->device));
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) { c->ipc_peer[r] = c->ipc_local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, all_handles + (size_t)64 * r, 64);
        hipError_t e = hipIpcOpenMemHandle(&c->ipc_peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            c->ipc_peer[r] = nullptr;
            return fail(c, MPST_ERR_DEVICE, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
        }
    }
    c->use_ipc = true;
    c->ipc_dead = false;
    c->epoch++;
    return 0;
}

int mpst_comm_select(void* ctx, int oneshot) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (oneshot && !c->ipc_peer[c->nranks > 1 ? (c->rank ^ 1) % c->nranks : 0]) return fail(c, MPST_ERR_INVALID, "inboxes are not attached");
    if (!oneshot && !c->comm) return fail(c, MPST_ERR_INVALID, "no RCCL communicator");
    c->use_ipc = oneshot != 0;
    c->epoch++;
    return 0;
}

int mpst_set_options(void* ctx, const mpst_options* o) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !o) return fail(c, MPST_ERR_INVALID, "NULL argument");
    if (o->chi_max < 1) return fail(c, MPST_ERR_INVALID, "chi_max must be >= 1");
    if (o->update_iters < 1) return fail(c, MPST_ERR_INVALID, "update_iters must be >= 1");
    if (o->loss != MPST_LOSS_KLD && o->loss != MPST_LOSS_MSE)
        return fail(c, MPST_ERR_UNSUPPORTED, "loss_grad must be KLD or MSE (Mixed exists only in the legacy ITensor path)");
    if (o->optimiser != MPST_OPT_TSGO && o->optimiser != MPST_OPT_GD)
        return fail(c, MPST_ERR_UNSUPPORTED, "Optim/OptimKit based solvers currently unimplemented for this version, set 'use_legacy_ITensor=true' in MPSOptions to enable");
    if (o->loss == MPST_LOSS_MSE && o->train_classes_separately)
        return fail(c, MPST_ERR_UNSUPPORTED, "no Loss_Grad_MSE method for TrainSeparate{true} (loss_functions.jl:561)");
    if (c->have_mps && o->chi_max > c->cap)
        return fail(c, MPST_ERR_INVALID, "chi_max %d exceeds the capacity %d fixed when the MPS was set; call mpst_set_options before mpst_set_mps", o->chi_max, c->cap);
    const bool resize = !c->have_opt || o->rescale_before != c->opt.rescale_before || o->update_iters != c->opt.update_iters;
    c->opt = *o;
    c->have_opt = true;
    c->epoch++;
    if (resize) c->ws_ready = false;
    return 0;
}

// Everything of a data set except the encoded values: validation, class counts, class-pure spans.
static int dataset_common(Ctx* c, int which, const int32_t* label_idx, int64_t N, int32_t T, int32_t d, int32_t C,
                          const int64_t* n_global_per_class, bool have_values) {
    if (which != MPST_TRAIN && which != MPST_TEST) return fail(c, MPST_ERR_INVALID, "which must be 0 (train) or 1 (test)");
    if (N < 0 || T < 2 || d < 1 || C < 1) return fail(c, MPST_ERR_INVALID, "bad data set dimensions");
    if ((c->T && c->T != T) || (c->d && c->d != d) || (c->C && c->C != C)) {
        if (c->have_mps || c->ds[which ^ 1].N > 0)
            return fail(c, MPST_ERR_INVALID, "data set dimensions (T=%d,d=%d,C=%d) disagree with the context (T=%d,d=%d,C=%d)", T, d, C, c->T, c->d, c->C);
    }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->T = T; c->d = d; c->C = C;
    DataSet& s = c->ds[which];
    free_dataset(s);
    if (which == MPST_TRAIN) {
        c->ws_ready = false;        // caches, yhat, partials are sized by the training set
        c->caches_valid = false;
    }
    c->eval_ready = false;
    c->epoch++;
    s.N = N;
    s.counts.assign(C, 0);
    if (N == 0) return 0;
    if (!have_values || !label_idx) return fail(c, MPST_ERR_INVALID, "NULL data pointer");
    for (int64_t i = 0; i < N; ++i) {
        const int32_t l = label_idx[i];
        if (l < 0 || l >= C) return fail(c, MPST_ERR_INVALID, "label_idx[%lld] = %d out of range", (long long)i, l);
        if (i && l < label_idx[i - 1]) return fail(c, MPST_ERR_INVALID, "Training data must be sorted by class!");  // :624
        s.counts[l]++;
    }
    s.gcounts.assign(C, 0);
    s.Nglobal = 0;
    for (int k = 0; k < C; ++k) {
        s.gcounts[k] = n_global_per_class ? n_global_per_class[k] : s.counts[k];
        s.Nglobal += s.gcounts[k];
    }
    int rc;
    if ((rc = dalloc_e(c, &s.phi, N * T * d))) return rc;
    if ((rc = dalloc(c, &s.label, N))) return rc;
    HIPC(c, hipMemcpy(s.label, label_idx, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
    // class-pure spans
    std::vector<Span> tiles, chunks;
    std::vector<int32_t> coff(C + 1, 0);
    int64_t start = 0;
    for (int k = 0; k < C; ++k) {
        coff[k] = (int32_t)chunks.size();
        for (int64_t o = 0; o < s.counts[k]; o += TILE_S)
            tiles.push_back({(int32_t)(start + o), (int32_t)std::min<int64_t>(TILE_S, s.counts[k] - o), k, 0});
        for (int64_t o = 0; o < s.counts[k]; o += CHUNK_S)
            chunks.push_back({(int32_t)(start + o), (int32_t)std::min<int64_t>(CHUNK_S, s.counts[k] - o), k, 0});
        start += s.counts[k];
    }
    coff[C] = (int32_t)chunks.size();
    s.ntiles = (int32_t)tiles.size();
    s.nchunks = (int32_t)chunks.size();
    if ((rc = dalloc(c, &s.tiles, (int64_t)tiles.size()))) return rc;
    HIPC(c, hipMemcpy(s.tiles, tiles.data(), tiles.size() * sizeof(Span), hipMemcpyHostToDevice));
    if ((rc = dalloc(c, &s.chunks, (int64_t)chunks.size()))) return rc;
    HIPC(c, hipMemcpy(s.chunks, chunks.data(), chunks.size() * sizeof(Span), hipMemcpyHostToDevice));
    if ((rc = dalloc(c, &s.cls_chunk_off, C + 1))) return rc;
    HIPC(c, hipMemcpy(s.cls_chunk_off, coff.data(), (size_t)(C + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> soff(C + 1, 0);
        for (int k = 0; k < C; ++k) soff[k + 1] = soff[k] + (int32_t)s.counts[k];
        if ((rc = dalloc(c, &s.cls_off, C + 1))) return rc;
        HIPC(c, hipMemcpy(s.cls_off, soff.data(), (size_t)(C + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // parts of the fused gradient kernel: class-pure runs of whole 16-series tiles, about PARTS_TARGET of them
    {
        int64_t tiles_total = 0;
        for (int k = 0; k < C; ++k) tiles_total += (s.counts[k] + TILE_S - 1) / TILE_S;
        int target = tiles_total >= 4 * PARTS_TARGET ? 2 * PARTS_TARGET : PARTS_TARGET;
        if (const char* e = getenv("MPST_PARTS")) target = std::max(1, atoi(e));
        for (int pk = 0; pk < 2; ++pk) {
            const int tgt = pk ? std::max(1, target / C) : target;      // MSE: every run is walked once per class
            std::vector<Part> runs;
            int64_t st = 0;
            for (int k = 0; k < C; ++k) {
                const int64_t tk = (s.counts[k] + TILE_S - 1) / TILE_S;
                if (tk > 0) {
                    int64_t pkn = (tgt * tk + tiles_total / 2) / std::max<int64_t>(tiles_total, 1);
                    pkn = std::max<int64_t>(1, std::min(pkn, tk));
                    for (int64_t q = 0; q < pkn; ++q) {
                        const int64_t t0 = tk * q / pkn, t1 = tk * (q + 1) / pkn;
                        const int64_t a = st + t0 * TILE_S, bnd = std::min(st + t1 * TILE_S, st + s.counts[k]);
                        runs.push_back({(int32_t)a, (int32_t)(bnd - a), k, k, 0, 0, 0, 0});
                    }
                }
                st += s.counts[k];
            }
            std::vector<Part> parts;
            std::vector<int32_t> poff(C + 1, 0);
            for (int cc = 0; cc < C; ++cc) {
                poff[cc] = (int32_t)parts.size();
                bool first = true;
                for (const Part& r : runs) {
                    if (!pk && r.own != cc) continue;
                    Part q = r;
                    q.cls = cc;
                    q.first_of_cls = first ? 1 : 0;
                    first = false;
                    parts.push_back(q);
                }
            }
            poff[C] = (int32_t)parts.size();
            s.nparts[pk] = (int32_t)parts.size();
            if ((rc = dalloc(c, &s.parts[pk], (int64_t)parts.size()))) return rc;
            if (!parts.empty()) HIPC(c, hipMemcpy(s.parts[pk], parts.data(), parts.size() * sizeof(Part), hipMemcpyHostToDevice));
            if ((rc = dalloc(c, &s.part_off[pk], C + 1))) return rc;
            HIPC(c, hipMemcpy(s.part_off[pk], poff.data(), (size_t)(C + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    std::vector<double> inv(C);
    for (int k = 0; k < C; ++k) inv[k] = s.gcounts[k] > 0 ? 1.0 / (double)s.gcounts[k] : 0.0;
    if ((rc = dalloc(c, &s.inv_count, C))) return rc;
    HIPC(c, hipMemcpy(s.inv_count, inv.data(), (size_t)C * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

// the element type of a context is fixed by its first data set (or mpst_set_dtype) and can only change once both are gone
static int set_ctx_dtype(Ctx* c, int which, int dtype) {
    if (dtype != MPST_F64 && dtype != MPST_F32 && dtype != MPST_C128 && dtype != MPST_C64) return fail(c, MPST_ERR_INVALID, "unknown dtype %d", dtype);
    const bool other = c->ds[which ^ 1].N > 0 || c->have_mps;
    if (c->have_dtype && dtype != c->dtype && other)
        return fail(c, MPST_ERR_INVALID, "dtype %d disagrees with the context's element type %d (data sets and MPS share one element type, opts.dtype)", dtype, c->dtype);
    if (!c->have_dtype || dtype != c->dtype) {
        c->ws_ready = false;
        c->eval_ready = false;
    }
    c->dtype = dtype;
    c->have_dtype = true;
    c->zw = (dtype == MPST_C128 || dtype == MPST_C64) ? 2 : 1;
    c->esz = (size_t)((dtype == MPST_F32 || dtype == MPST_C64) ? 4 : 8) * c->zw;
    c->typed = dtype != MPST_F64 || getenv("MPST_TYPED") != nullptr;
    return 0;
}

int mpst_set_dtype(void* ctx, int32_t dtype) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (c->ds[0].N > 0 || c->ds[1].N > 0 || c->have_mps) {
        if (c->have_dtype && dtype == c->dtype) return 0;
        return fail(c, MPST_ERR_INVALID, "mpst_set_dtype must precede the data sets and the MPS");
    }
    return set_ctx_dtype(c, 0, dtype);
}

int mpst_set_dataset(void* ctx, int which, const void* phi_, const int32_t* label_idx, int64_t N, int32_t T, int32_t d,
                     int32_t C, int32_t dtype, const int64_t* n_global_per_class) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (which != MPST_TRAIN && which != MPST_TEST) return fail(c, MPST_ERR_INVALID, "which must be 0 (train) or 1 (test)");
    int rc = set_ctx_dtype(c, which, dtype);
    if (rc) return rc;
    rc = dataset_common(c, which, label_idx, N, T, d, C, n_global_per_class, phi_ != nullptr);
    if (rc || N == 0) return rc;
    DataSet& s = c->ds[which];
    // site-major copy [T][N][d]
    const size_t row = (size_t)d * c->esz;
    const char* phi = (const char*)phi_;
    std::vector<char> tmp((size_t)N * T * row);
    for (int64_t i = 0; i < N; ++i)
        for (int t = 0; t < T; ++t) memcpy(&tmp[((size_t)t * N + i) * row], &phi[((size_t)i * T + t) * row], row);
    HIPC(c, hipMemcpy(s.phi, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    return 0;
}

// preprocessing + encoding of X[N][T] into dphi ([T][N][d] doubles, or (re, im) pairs for the Fourier basis): shared by
// mpst_encode_dataset (dphi = the data set's product states) and mpst_encode_values (dphi = scratch, copied out)
static int encode_core(Ctx* c, const double* X, int64_t N, int32_t T, int32_t d, mpst_encode_opts* eo, double* dphi, double* oob_fix,
                       double* seconds) {
    int rc;
    if (!eo) return fail(c, MPST_ERR_INVALID, "NULL encode options");
    const bool fit_sig = eo->sigmoid_transform && eo->fit_sigmoid && !eo->is_test;
    if (eo->fit_sigmoid && eo->is_test) return fail(c, MPST_ERR_INVALID, "fit_sigmoid: the RobustSigmoid is fitted on the training set only");
    if (eo->sigmoid_transform && !fit_sig && !(eo->iqr > 0.0)) return fail(c, MPST_ERR_INVALID, "robust sigmoid needs iqr > 0");
    if (fit_sig && N * (int64_t)T > 0x7fffffffll) return fail(c, MPST_ERR_UNSUPPORTED, "fit_sigmoid sorts at most 2^31 - 1 values");
    double *dX = nullptr, *part = nullptr, *lohi = nullptr, *fix = nullptr;
    struct Temps {      // released on every exit path, the HIPC early returns included
        double **a, **b, **cc, **d;
        ~Temps() { dfree(a); dfree(b); dfree(cc); dfree(d); }
    } temps{&dX, &part, &lohi, &fix};
    if ((rc = dalloc(c, &dX, N * T)) || (rc = dalloc(c, &part, 512)) || (rc = dalloc(c, &lohi, 2))) return rc;
    const bool test = eo->is_test != 0;
    if (test && eo->rescale_out_of_bounds && (rc = dalloc(c, &fix, 2 * N))) return rc;
    HIPC(c, hipMemcpy(dX, X, (size_t)N * T * sizeof(double), hipMemcpyHostToDevice));
    double fit_seconds = 0.0;
    if (fit_sig) {
        // Normalization.fit(RobustSigmoid, X_train) (utils.jl:174): median and quartiles of all values from a device sort
        double *sorted = nullptr, *q3 = nullptr;
        uint8_t* tmp = nullptr;
        struct T2 {
            double **a, **b; uint8_t** t;
            ~T2() { dfree(a); dfree(b); dfree(t); }
        } t2{&sorted, &q3, &tmp};
        const size_t tb = order_stats_temp_bytes(N * T);
        if ((rc = dalloc(c, &sorted, N * T)) || (rc = dalloc(c, &q3, 3)) || (rc = dalloc(c, &tmp, (int64_t)tb))) return rc;
        HIPC(c, hipEventRecord(c->ev_start, c->stream));
        HIPC(c, launch_order_stats(dX, sorted, tmp, tb, N * T, q3, c->stream));
        HIPC(c, hipEventRecord(c->ev_stop, c->stream));
        HIPC(c, hipEventSynchronize(c->ev_stop));
        float fms = 0.f;
        HIPC(c, hipEventElapsedTime(&fms, c->ev_start, c->ev_stop));
        fit_seconds = 1e-3 * fms;
        double h3[3];
        HIPC(c, hipMemcpy(h3, q3, sizeof h3, hipMemcpyDeviceToHost));
        eo->median = h3[0];
        eo->iqr = h3[2] - h3[1];
        if (!(eo->iqr > 0.0)) return fail(c, MPST_ERR_INVALID, "robust sigmoid needs iqr > 0 (the training data has iqr = %g)", eo->iqr);
    }
    if (test || !eo->minmax) {
        const double h[2] = {eo->lo, eo->hi};
        HIPC(c, hipMemcpy(lohi, h, sizeof h, hipMemcpyHostToDevice));
    }
    EncDev e{};
    e.N = N; e.T = T; e.d = d;
    e.norm = eo->basis == MPST_BASIS_LEGENDRE;
    e.fourier = eo->basis == MPST_BASIS_FOURIER || eo->basis == MPST_BASIS_STOUDENMIRE || eo->basis == MPST_BASIS_SAHAND;
    e.basis = eo->basis;
    e.sigmoid = eo->sigmoid_transform; e.minmax = eo->minmax; e.is_test = test;
    e.med = eo->median; e.s = eo->iqr / 1.35;
    e.lb = eo->data_lb; e.ub = eo->data_ub; e.a = eo->range_a; e.b = eo->range_b;
    e.nrm = std::sqrt(std::sqrt((2 * d + 1) / 2.0) * d);
    e.lohi = lohi; e.fix = fix;
    HIPC(c, hipEventRecord(c->ev_start, c->stream));
    launch_encode(e, dX, dphi, part, lohi, fix, !test && eo->minmax, c->stream);
    HIPC(c, hipEventRecord(c->ev_stop, c->stream));
    HIPC(c, hipEventSynchronize(c->ev_stop));
    HIPC(c, hipGetLastError());
    float ms = 0.f;
    HIPC(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    if (seconds) *seconds = 1e-3 * ms + fit_seconds;
    if (!test && eo->minmax) {
        double h[2];
        HIPC(c, hipMemcpy(h, lohi, sizeof h, hipMemcpyDeviceToHost));
        eo->lo = h[0];
        eo->hi = h[1];
    }
    if (fix && oob_fix) HIPC(c, hipMemcpy(oob_fix, fix, (size_t)2 * N * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int mpst_encode_dataset(void* ctx, int which, const double* X, const int32_t* label_idx, int64_t N, int32_t T, int32_t d,
                        int32_t C, mpst_encode_opts* eo, const int64_t* n_global_per_class, double* oob_fix, double* seconds) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (!eo) return fail(c, MPST_ERR_INVALID, "NULL encode options");
    if (eo->basis < MPST_BASIS_LEGENDRE || eo->basis > MPST_BASIS_UNIFORM)
        return fail(c, MPST_ERR_UNSUPPORTED, "device-side encoding implements the closed-form bases (Legendre, Fourier, Stoudenmire, Sahand, Uniform)");
    if (eo->basis == MPST_BASIS_STOUDENMIRE && d != 2) return fail(c, MPST_ERR_INVALID, "Stoudenmire Angle encoding only supports d = 2!");
    if (eo->basis == MPST_BASIS_SAHAND && d % 2) return fail(c, MPST_ERR_INVALID, "Sahand encoding only supports even dimension");
    if (which != MPST_TRAIN && which != MPST_TEST) return fail(c, MPST_ERR_INVALID, "which must be 0 (train) or 1 (test)");
    // the element type: what mpst_set_dtype / the other data set fixed (opts.dtype), else the basis' own (Float64 / ComplexF64)
    const bool bcx = eo->basis == MPST_BASIS_FOURIER || eo->basis == MPST_BASIS_STOUDENMIRE || eo->basis == MPST_BASIS_SAHAND;
    const int natural = bcx ? MPST_C128 : MPST_F64;
    const int want = c->have_dtype ? c->dtype : natural;
    if (bcx && (want == MPST_F64 || want == MPST_F32))
        return fail(c, MPST_ERR_INVALID, "Using a complex valued encoding but the MPS is real. If using a complex-valued custom encoding, set 'dtype <: Complex' in MPSOptions");   // RealRealHighDimension.jl:462-464
    int rc = set_ctx_dtype(c, which, want);
    if (rc) return rc;
    rc = dataset_common(c, which, label_idx, N, T, d, C, n_global_per_class, X != nullptr);
    if (rc || N == 0) return rc;
    if (want == natural) return encode_core(c, X, N, T, d, eo, c->ds[which].phi, oob_fix, seconds);
    double* tmp = nullptr;
    struct T1 {
        double** a;
        ~T1() { dfree(a); }
    } t1{&tmp};
    if ((rc = dalloc(c, &tmp, N * T * d * (bcx ? 2 : 1)))) return rc;
    if ((rc = encode_core(c, X, N, T, d, eo, tmp, oob_fix, seconds))) return rc;
    launch_tcast(tmp, bcx ? 1 : 0, c->ds[which].phi, c->zw == 2, want == MPST_F32 || want == MPST_C64, N * T * d, c->stream);
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int mpst_encode_values(void* ctx, const double* X, int64_t N, int32_t T, int32_t d, mpst_encode_opts* eo, void* phi_out, double* oob_fix,
                       double* seconds) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (!eo || !X || !phi_out) return fail(c, MPST_ERR_INVALID, "NULL argument");
    if (N <= 0 || T < 1 || d < 1 || d > 64) return fail(c, MPST_ERR_INVALID, "bad dimensions");
    if (eo->basis < MPST_BASIS_LEGENDRE || eo->basis > MPST_BASIS_UNIFORM)
        return fail(c, MPST_ERR_UNSUPPORTED, "device-side encoding implements the closed-form bases (Legendre, Fourier, Stoudenmire, Sahand, Uniform)");
    if (eo->basis == MPST_BASIS_STOUDENMIRE && d != 2) return fail(c, MPST_ERR_INVALID, "Stoudenmire Angle encoding only supports d = 2!");
    if (eo->basis == MPST_BASIS_SAHAND && d % 2) return fail(c, MPST_ERR_INVALID, "Sahand encoding only supports even dimension");
    HIPC(c, hipSetDevice(c->device));
    const int zw = (eo->basis == MPST_BASIS_FOURIER || eo->basis == MPST_BASIS_STOUDENMIRE || eo->basis == MPST_BASIS_SAHAND) ? 2 : 1;
    double* dphi = nullptr;
    struct T1 {
        double** a;
        ~T1() { dfree(a); }
    } t1{&dphi};
    int rc;
    if ((rc = dalloc(c, &dphi, N * T * d * zw))) return rc;
    if ((rc = encode_core(c, X, N, T, d, eo, dphi, oob_fix, seconds))) return rc;
    std::vector<double> tmp((size_t)N * T * d * zw);
    HIPC(c, hipMemcpy(tmp.data(), dphi, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    double* out = (double*)phi_out;
    const size_t w = (size_t)d * zw;
    for (int64_t i = 0; i < N; ++i)
        for (int t = 0; t < T; ++t) memcpy(&out[((size_t)i * T + t) * w], &tmp[((size_t)t * N + i) * w], w * sizeof(double));
    return 0;
}

int mpst_get_encoded(void* ctx, int which, double* phi_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !phi_out) return MPST_ERR_INVALID;
    if (which != MPST_TRAIN && which != MPST_TEST) return fail(c, MPST_ERR_INVALID, "which must be 0 (train) or 1 (test)");
    const DataSet& s = c->ds[which];
    if (s.N == 0) return 0;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const int64_t N = s.N;
    const int T = c->T;
    const size_t row = (size_t)c->d * c->esz;             // phi_out holds elements of the context's type
    std::vector<char> tmp((size_t)N * T * row);
    HIPC(c, hipMemcpy(tmp.data(), s.phi, tmp.size(), hipMemcpyDeviceToHost));
    char* out = (char*)phi_out;
    for (int64_t i = 0; i < N; ++i)
        for (int t = 0; t < T; ++t) memcpy(&out[((size_t)i * T + t) * row], &tmp[((size_t)t * N + i) * row], row);
    return 0;
}

int mpst_set_mps(void* ctx, const void* const* site, const int32_t* chi, int32_t T, int32_t label_site) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !site || !chi) return fail(c, MPST_ERR_INVALID, "NULL argument");
    if (!c->have_opt) return fail(c, MPST_ERR_INVALID, "call mpst_set_options before mpst_set_mps");
    if (c->T == 0) return fail(c, MPST_ERR_INVALID, "call mpst_set_dataset before mpst_set_mps");
    if (T != c->T) return fail(c, MPST_ERR_INVALID, "MPS has %d sites, data has %d", T, c->T);
    if (label_site < 0 || label_site >= T) return fail(c, MPST_ERR_INVALID, "label_site out of range");
    if (chi[0] != 1 || chi[T] != 1) return fail(c, MPST_ERR_INVALID, "chi[0] and chi[T] must be 1");
    int cap = c->opt.chi_max;
    for (int j = 0; j <= T; ++j) {
        if (chi[j] < 1) return fail(c, MPST_ERR_INVALID, "chi[%d] < 1", j);
        cap = std::max(cap, (int)chi[j]);
    }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    const int d = c->d, C = c->C;
    if (cap != c->cap || !c->sites) {
        c->cap = cap;
        c->site_stride = (int64_t)C * cap * d * cap;
        int rc;
        if ((rc = dalloc_e(c, &c->sites, c->site_stride * T))) return rc;
        if ((rc = dalloc(c, &c->chi, T + 1))) return rc;
        if ((rc = dalloc(c, &c->label_site, 1))) return rc;
        c->ws_ready = false;
    }
    // boundary layout (s, l, r[, c]) column-major  ->  internal [c][l][s][r]; elements of esz bytes (a pure permutation)
    const size_t esz = c->esz;
    std::vector<char> buf((size_t)c->site_stride * T * esz, 0);
    for (int j = 0; j < T; ++j) {
        const int Dl = chi[j], Dr = chi[j + 1], Cj = (j == label_site) ? C : 1;
        const char* src = (const char*)site[j];
        if (!src) return fail(c, MPST_ERR_INVALID, "site[%d] is NULL", j);
        char* dst = &buf[(size_t)j * c->site_stride * esz];
        for (int cc = 0; cc < Cj; ++cc)
            for (int r = 0; r < Dr; ++r)
                for (int l = 0; l < Dl; ++l)
                    for (int s = 0; s < d; ++s)
                        memcpy(dst + ((((size_t)cc * Dl + l) * d + s) * Dr + r) * esz, src + (s + (size_t)d * (l + (size_t)Dl * (r + (size_t)Dr * cc))) * esz, esz);
    }
    HIPC(c, hipMemcpy(c->sites, buf.data(), buf.size(), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->chi, chi, (size_t)(T + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->label_site, &label_site, sizeof(int32_t), hipMemcpyHostToDevice));
    c->have_mps = true;
    c->caches_valid = false;
    c->host_label_site = label_site;
    c->epoch++;
    return 0;
}

int mpst_get_chi(void* ctx, int32_t* chi_out, int32_t* label_site) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->have_mps) return fail(c, MPST_ERR_INVALID, "no MPS set");
    HIPC(c, hipSetDevice(c->device));
    std::vector<int32_t> chi; int32_t ls;
    int rc = host_chi(c, chi, &ls);
    if (rc) return rc;
    if (chi_out) memcpy(chi_out, chi.data(), chi.size() * sizeof(int32_t));
    if (label_site) *label_site = ls;
    return 0;
}

int mpst_get_mps(void* ctx, void* const* site_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->have_mps || !site_out) return fail(c, MPST_ERR_INVALID, "no MPS set / NULL argument");
    HIPC(c, hipSetDevice(c->device));
    std::vector<int32_t> chi; int32_t ls;
    int rc = host_chi(c, chi, &ls);
    if (rc) return rc;
    const size_t esz = c->esz;
    std::vector<char> buf((size_t)c->site_stride * c->T * esz);
    HIPC(c, hipMemcpy(buf.data(), c->sites, buf.size(), hipMemcpyDeviceToHost));
    const int d = c->d;
    for (int j = 0; j < c->T; ++j) {
        const int Dl = chi[j], Dr = chi[j + 1], Cj = (j == ls) ? c->C : 1;
        char* dst = (char*)site_out[j];
        if (!dst) return fail(c, MPST_ERR_INVALID, "site_out[%d] is NULL", j);
        const char* src = &buf[(size_t)j * c->site_stride * esz];
        for (int cc = 0; cc < Cj; ++cc)
            for (int r = 0; r < Dr; ++r)
                for (int l = 0; l < Dl; ++l)
                    for (int s = 0; s < d; ++s)
                        memcpy(dst + (s + (size_t)d * (l + (size_t)Dl * (r + (size_t)Dr * cc))) * esz, src + ((((size_t)cc * Dl + l) * d + s) * Dr + r) * esz, esz);
    }
    return 0;
}

int mpst_build_caches(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    const int32_t ls = c->host_label_site;
    // environments on both sides of the label site p: LE[0..p-1] and RE[T-1..p+1].  With the label
    // on the last site (the state fitMPS starts from) this is construct_caches(W; going_left=true).
    View v = make_view(c, MPST_TRAIN);
    const int64_t cs = (int64_t)v.N * v.cap;
    if (c->typed) {
        enqueue_caches_typed(c, ls, ls);
    } else if (env_walk_on(c, v)) {
        ProfScope p(c, K_ENV);
        launch_env_walk(v, 1, std::min(ls, c->T - 1), c->stream);
        launch_env_walk(v, 0, c->T - 1 - ls, c->stream);
    } else {
        ProfScope p(c, K_ENV);
        for (int j = 0; j < ls && j <= c->T - 2; ++j)
            launch_env(v, j, 1, j > 0 ? c->LE + (int64_t)(j - 1) * cs : nullptr, j, ENV_M_SITE, j + 1,
                       c->LE + (int64_t)j * cs, c->stream);
        for (int j = c->T - 1; j > ls && j >= 1; --j)
            launch_env(v, j, 0, j < c->T - 1 ? c->RE + (int64_t)(j + 1) * cs : nullptr, j + 1, ENV_M_SITE_T, j,
                       c->RE + (int64_t)j * cs, c->stream);
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    c->caches_valid = true;
    c->ynext_lid = -1;
    return 0;
}

int mpst_sweep(void* ctx, mpst_sweep_stats* out) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    if (!c->caches_valid)
        return fail(c, MPST_ERR_INVALID, "the environment caches do not describe the current MPS / data set: call mpst_build_caches first");
    if (c->host_label_site != c->T - 1)
        return fail(c, MPST_ERR_INVALID, "a sweep starts with the label index on the last site (RealRealHighDimension.jl:19-29), it is on site %d", c->host_label_site);
    View v = make_view(c, MPST_TRAIN);
    // k0 > 0: the rest of a sweep from its k0-th bond on (the state is that after bond k0 - 1: mpst_sweep's tail recovery)
    auto enqueue_sweep = [&](int k0 = 0) -> int {
        int r0 = enqueue_reset_status(c);
        if (r0) return r0;
        c->ynext_lid = -1;
        // bond order of one sweep (:731, :776); unless the tensor is rescaled first or the caches are
        // rebuilt in between, bond k+1's tensor is assembled by bond k's environment kernel
        const int nb = c->T - 1;
        const bool chain = !v.rescale_before;
        int r;
        for (int k = k0; k < 2 * nb; ++k) {
            const int lid = k < nb ? nb - 1 - k : k - nb, left = k < nb;
            const bool boundary_before = c->opt.rebuild_caches && (k == nb);
            const bool boundary_after = c->opt.rebuild_caches && (k == nb - 1);
            bool have = chain && k > k0 && !boundary_before;
            if (c->fused && k == nb) have = false;      // turning point: the same bond again, nothing was chained
            int next = -1;
            if (chain && k + 1 < 2 * nb && !boundary_after) next = (k + 1) < nb ? nb - 1 - (k + 1) : (k + 1) - nb;
            if ((r = enqueue_bond(c, v, lid, left, have, next, k))) return r;
            if (c->opt.rebuild_caches && k == nb - 1) enqueue_caches(c, v, 0);      // :770
        }
        if (c->opt.rebuild_caches) enqueue_caches(c, v, 1);                         // :804
        return 0;
    };
    // The 2(T-1) x 11 launches of a sweep are replayed from a hipGraph: nothing in the sequence depends
    // on host-side state (bond dimensions are read on the device), and the pre-built dispatch packets
    // shorten the dependent kernel-to-kernel hand-over that dominates the small kernels.  Per-kernel
    // profiling and the RCCL leg keep the plain stream path.
    // (large bonds stay on the plain stream: replaying their ~8000 launches from a graph was measured to gain nothing - 725.0
    // against 724.5 ms per sweep at (8192, 200, 64, 8) - and a capture must not overlap other threads' legacy-stream copies)
    const bool use_graph = !multi(c) && !c->big && c->prof_mask == 0 && getenv("MPST_NO_GRAPH") == nullptr;
    if (use_graph && (!c->sweep_graph || c->graph_epoch != c->epoch)) {
        if (c->sweep_graph) {
            (void)hipGraphExecDestroy(c->sweep_graph);
            c->sweep_graph = nullptr;
        }
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        rc = enqueue_sweep();
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess || !g) return fail(c, MPST_ERR_DEVICE, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        HIPC(c, hipGetLastError());     // a launch rejected during capture (bad configuration) surfaces here
        e = hipGraphInstantiate(&c->sweep_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            c->sweep_graph = nullptr;
            return fail(c, MPST_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        }
        c->graph_epoch = c->epoch;
    }
    // large bonds: no verdict is read inside the sweep (launch_eig_blocked_nosync); the state the sweep starts from is kept
    // so that a sweep in which a bond failed can be redone bond by bond
    // (one rank only: a persistent kernel that runs out of patience is a rank-local event, and a rank that redoes its sweep
    // alone would issue all-reduces its peers do not)
    const bool optimistic = c->big && c->blk && c->big_opt && !multi(c) && c->big_cooldown == 0;
    if (c->big_cooldown > 0) c->big_cooldown--;
    const size_t site_bytes = (size_t)c->site_stride * c->T * c->esz, chi_bytes = (size_t)(c->T + 1) * sizeof(int32_t);
    if (optimistic) {
        if (!c->snap_sites) {
            // all three or none: a partial failure must not leave a half-allocated snapshot behind
            if ((rc = dalloc_e(c, &c->snap_sites, c->site_stride * c->T)) || (rc = dalloc(c, &c->snap_chi, c->T + 2)) || (rc = dalloc(c, &c->snap_sc, 1))) {
                dfree(&c->snap_sites); dfree(&c->snap_chi); dfree(&c->snap_sc);
                return rc;
            }
        }
        HIPC(c, hipMemcpyAsync(c->snap_sites, c->sites, site_bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->snap_chi, c->chi, chi_bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->snap_chi + c->T + 1, c->label_site, sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
        HIPC(c, hipMemcpyAsync(c->snap_sc, c->sc, sizeof(DevScalars), hipMemcpyDeviceToDevice, c->stream));
        c->big_opt_active = true;
    }
    struct ActiveGuard {        // whatever path leaves this function, the per-bond eigensolver path reads its verdict again
        bool& f;
        ~ActiveGuard() { f = false; }
    } active_guard{c->big_opt_active};
    HIPC(c, hipEventRecord(c->ev_start, c->stream));
    if (use_graph) {
        HIPC(c, hipGraphLaunch(c->sweep_graph, c->stream));
    } else if ((rc = enqueue_sweep())) {
        c->big_opt_active = false;
        return rc;
    }
    HIPC(c, hipGetLastError());         // launch-time failures of the ~2000 enqueues above
    c->ynext_lid = -1;
    if (optimistic) {
        c->big_opt_active = false;
        const int st = blocked_eig_take_sticky(c->blk, c->stream);        // the one synchronisation of the sweep
        if (st < 0) return fail(c, MPST_ERR_DEVICE, "reading the sweep's eigensolver verdict failed");
        if (st) {
            // some bond's verification failed, or a persistent tridiagonalisation gave up: everything after it ran on
            // unspecified data.  Back to the start of the sweep, caches rebuilt, bond by bond with the verdict read each time.
            c->big_redos++;
            c->big_cooldown = 4;        // whatever made the persistent kernels fail (a shared GPU, CU masking) tends to last: the next sweeps take the per-bond path
            HIPC(c, hipMemcpyAsync(c->sites, c->snap_sites, site_bytes, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(c->chi, c->snap_chi, chi_bytes, hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(c->label_site, c->snap_chi + c->T + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            HIPC(c, hipMemcpyAsync(c->sc, c->snap_sc, sizeof(DevScalars), hipMemcpyDeviceToDevice, c->stream));
            enqueue_caches(c, v, 1);
            if ((rc = enqueue_sweep())) return rc;
            HIPC(c, hipGetLastError());
        }
    }
    HIPC(c, hipEventRecord(c->ev_stop, c->stream));
    HIPC(c, hipEventSynchronize(c->ev_stop));
    float ms = 0.f;
    HIPC(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    prof_collect(c);
    DevScalars sc;
    HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    int tail_fallbacks = 0;
    if (sc.redo > 0 && !c->chain4_hold) {
        // four-launch chain: a tail launch whose verification failed (clustered kept eigenvalues: the case k_eig_fin hands to its Jacobi
        // solver) has marked the sweep and left the MPS, the caches and the chained tensor as the bond before it left them; so did every
        // tail launch after it.  The rest of the sweep runs on the six-launch chain, plain stream.
        c->tail_redos++;
        tail_fallbacks = sc.eig_fallbacks;
        c->chain4_hold = true;
        struct Hold { bool& h; ~Hold() { h = false; } } hold{c->chain4_hold};
        v = make_view(c, MPST_TRAIN);
        HIPC(c, hipEventRecord(c->ev_start, c->stream));
        if ((rc = enqueue_sweep(sc.redo - 1))) return rc;
        HIPC(c, hipGetLastError());
        HIPC(c, hipEventRecord(c->ev_stop, c->stream));
        HIPC(c, hipEventSynchronize(c->ev_stop));
        float ms2 = 0.f;
        HIPC(c, hipEventElapsedTime(&ms2, c->ev_start, c->ev_stop));
        ms += ms2;
        prof_collect(c);
        HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
        sc.eig_fallbacks += tail_fallbacks;
    }
    refresh_ss_counts(c);
    std::vector<int32_t> chi(c->T + 1);
    HIPC(c, hipMemcpy(chi.data(), c->chi, chi.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out) {
        out->seconds = 1e-3 * ms;
        out->svd_status = sc.status;
        out->max_chi = *std::max_element(chi.begin(), chi.end());
        out->eig_sweeps_total = sc.eig_sweeps_total;
        out->eig_fallbacks = sc.eig_fallbacks;
    }
    if (sc.status) {
        c->caches_valid = false;        // the state after a failed bond is unspecified: set_mps + build_caches to go on
        if (sc.status == MPST_ERR_DEVICE && c->use_ipc) {
            c->use_ipc = false;         // flags / epochs / slots are in an undefined cross-rank state: never reuse them
            c->ipc_dead = true;
            return fail(c, MPST_ERR_DEVICE, "the one-shot all-reduce timed out waiting for a peer (MPST_AR_TIMEOUT_S): rank %d of %d saw rank %d's flag at epoch %d while waiting "
                        "for one of the %llu calls issued so far; the path is retired until the inboxes are exported again", c->rank, c->nranks, sc.pad[0], sc.pad[1],
                        (unsigned long long)c->ar_epoch);
        }
        return fail(c, MPST_ERR_SVD, "bond-tensor decomposition failed (non-finite spectrum or eigensolver did not converge)");
    }
    return 0;
}

// K independent fits of one shape advanced by ONE launch chain (hyper-parameter candidates, CV folds, restarts: what the
// reference farms out with @distributed, tuning.jl).  Every launch of the headline chain carries all K fits (blockIdx.z), so the
// kernel count per sweep is that of one fit; results are those of K separate mpst_sweep calls, bit for bit.
// phase 1: validation and (if the batch is new) capture of its graph; phase 2: replay and read-back; 0: both.  mpst_sweep_batch_multi
// prepares all its groups on the calling thread before any group runs: a capture in one thread is invalidated by another thread's
// synchronous copies (the read-back of a group that has finished its sweep) - "operation failed due to a previous error during
// capture", one run in a dozen when two groups captured side by side.
static int sweep_batch_impl(void* const* ctxs, int32_t K, mpst_sweep_stats* out, int phase) {
    if (!ctxs || K < 1 || K > 64) return fail(nullptr, MPST_ERR_INVALID, "mpst_sweep_batch: 1..64 contexts");
    Ctx* c0 = (Ctx*)ctxs[0];
    if (!c0) return MPST_ERR_INVALID;
    if (hipSetDevice(c0->device) != hipSuccess) return fail(c0, MPST_ERR_DEVICE, "cannot select device %d", c0->device);
    int rc;
    for (int k = 0; k < K; ++k) {
        Ctx* c = (Ctx*)ctxs[k];
        if (!c) return fail(c0, MPST_ERR_INVALID, "context %d is NULL", k);
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(c0, MPST_ERR_INVALID, "context %d appears twice", k);
        if ((rc = check_ready(c))) {
            if (c != c0) c0->err = c->err;
            return rc;
        }
        if (!c->caches_valid) return fail(c0, MPST_ERR_INVALID, "context %d: call mpst_build_caches first", k);
        if (c->host_label_site != c->T - 1) return fail(c0, MPST_ERR_INVALID, "context %d: a sweep starts with the label index on the last site", k);
        const bool chain_ok = c->fused && c->b2 && !c->typed && !multi(c) && eig_merged() && c->opt.update_iters == 1 && !c->opt.track_cost &&
                              !c->opt.rebuild_caches && c->prof_mask == 0;
        if (!chain_ok)
            return fail(c0, MPST_ERR_UNSUPPORTED, "context %d: mpst_sweep_batch runs the headline chain only (Float64, d*chi_max <= 128, <= 8192 series, one rank, "
                                                 "update_iters = 1, no track_cost / rebuild_caches / profiling)", k);
        const DataSet &a = c0->ds[MPST_TRAIN], &b = c->ds[MPST_TRAIN];
        const bool same = c->device == c0->device && c->T == c0->T && c->d == c0->d && c->C == c0->C && c->cap == c0->cap && a.N == b.N && a.ntiles == b.ntiles &&
                          a.counts == b.counts && c->opt.loss == c0->opt.loss && c->opt.train_classes_separately == c0->opt.train_classes_separately &&
                          c->opt.chi_max == c0->opt.chi_max && c->b2_ksplit == c0->b2_ksplit && (c->batch_hint > 1) == (c0->batch_hint > 1) && c->b2_norm_parts == c0->b2_norm_parts;
        if (!same) return fail(c0, MPST_ERR_UNSUPPORTED, "context %d differs in shape from context 0 (T, d, C, capacity, chi_max, class counts, loss): batch fits of one shape", k);
        HIPC(c0, hipStreamSynchronize(c->stream));
    }
    HIPC(c0, hipSetDevice(c0->device));
    if (c0->batch_cap < K) {
        dfree(&c0->batch_views);
        if ((rc = dalloc(c0, &c0->batch_views, (int64_t)2 * K))) return rc;
        c0->batch_cap = K;
        c0->batch_key.clear();
    }
    std::vector<std::pair<uint64_t, uint64_t>> key;
    for (int k = 0; k < K; ++k) key.push_back({((Ctx*)ctxs[k])->uid, ((Ctx*)ctxs[k])->epoch});
    const View v0 = make_view(c0, MPST_TRAIN);
    if (!c0->batch_graph || key != c0->batch_key) {
        if (c0->batch_graph) {
            (void)hipGraphExecDestroy(c0->batch_graph);
            c0->batch_graph = nullptr;
        }
        std::vector<View> hv((size_t)2 * c0->batch_cap);
        for (int k = 0; k < K; ++k) {
            Ctx* c = (Ctx*)ctxs[k];
            hv[k] = make_view(c, MPST_TRAIN);
            View vg = hv[k];                    // the Gram launch reads the loss pieces and the gradient-norm pieces k_grad_s left
            vg.n_lossp = c->b2_ksplit;
            vg.n_norm_part = c->b2_norm_parts;
            hv[(size_t)c0->batch_cap + k] = vg;
        }
        HIPC(c0, hipMemcpy(c0->batch_views, hv.data(), hv.size() * sizeof(View), hipMemcpyHostToDevice));
        const View* dv = c0->batch_views;
        const View* dvg = c0->batch_views + c0->batch_cap;
        hipStream_t s = c0->stream;
        HIPC(c0, hipStreamSynchronize(s));
        HIPC(c0, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int bad = 0;
        for (int k = 0; k < K && !bad; ++k)
            if (hipMemsetAsync((char*)((Ctx*)ctxs[k])->sc + offsetof(DevScalars, status), 0, 12, s) != hipSuccess) bad = 1;
        const int nb = c0->T - 1;
        const int64_t cs = (int64_t)v0.N * v0.cap;
        for (int q = 0; q < 2 * nb && !bad; ++q) {
            const int lid = q < nb ? nb - 1 - q : q - nb, left = q < nb, rid = lid + 1;
            const bool have = q > 0 && q != nb;                        // as mpst_sweep's fused chain: nothing was chained at the turning point
            const int next = q + 1 < 2 * nb ? ((q + 1) < nb ? nb - 1 - (q + 1) : (q + 1) - nb) : -1;
            const int chain = (next >= 0 && next == (left ? lid - 1 : lid + 1)) ? 1 : 0;
            if (!have) launch_bt_assemble_b(v0, dv, K, lid, s);
            launch_yhat_s_b(v0, dv, K, lid, s);
            launch_grad_s_b(v0, dv, K, lid, s);
            launch_gram_upd_b(v0, dvg, K, lid, left, 1, s);
            launch_eig_b(v0, dv, K, lid, left, 0, s);
            launch_eig_b(v0, dv, K, lid, left, 2, s);
            if (left) launch_env_split_b(v0, dv, K, lid, 1, rid, 0, rid < c0->T - 1 ? (int64_t)(rid + 1) * cs : -1, rid + 1, rid, (int64_t)rid * cs, chain, s);
            else launch_env_split_b(v0, dv, K, lid, 0, lid, 1, lid > 0 ? (int64_t)(lid - 1) * cs : -1, lid, lid + 1, (int64_t)lid * cs, chain, s);
        }
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(s, &g);
        if (bad || e != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            return fail(c0, MPST_ERR_DEVICE, "capture of the batched sweep failed: %s", hipGetErrorString(e));
        }
        HIPC(c0, hipGetLastError());
        e = hipGraphInstantiate(&c0->batch_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            c0->batch_graph = nullptr;
            return fail(c0, MPST_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        }
        c0->batch_key = key;
    }
    if (phase == 1) return 0;
    HIPC(c0, hipEventRecord(c0->ev_start, c0->stream));
    HIPC(c0, hipGraphLaunch(c0->batch_graph, c0->stream));
    HIPC(c0, hipEventRecord(c0->ev_stop, c0->stream));
    HIPC(c0, hipEventSynchronize(c0->ev_stop));
    HIPC(c0, hipGetLastError());
    float ms = 0.f;
    HIPC(c0, hipEventElapsedTime(&ms, c0->ev_start, c0->ev_stop));
    int failed = -1;
    for (int k = 0; k < K; ++k) {
        Ctx* c = (Ctx*)ctxs[k];
        DevScalars sc;
        HIPC(c0, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
        refresh_ss_counts(c);
        std::vector<int32_t> chi(c->T + 1);
        HIPC(c0, hipMemcpy(chi.data(), c->chi, chi.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (out) {
            out[k].seconds = 1e-3 * ms;            // of the whole batch: the fits advance together
            out[k].svd_status = sc.status;
            out[k].max_chi = *std::max_element(chi.begin(), chi.end());
            out[k].eig_sweeps_total = sc.eig_sweeps_total;
            out[k].eig_fallbacks = sc.eig_fallbacks;
        }
        if (sc.status) {
            c->caches_valid = false;
            if (failed < 0) failed = k;
        }
    }
    if (failed >= 0) return fail(c0, MPST_ERR_SVD, "bond-tensor decomposition failed in fit %d of the batch (its svd_status is set; the other fits are intact)", failed);
    return 0;
}

int mpst_sweep_batch(void* const* ctxs, int32_t K, mpst_sweep_stats* out) { return sweep_batch_impl(ctxs, K, out, 0); }

// K fits dealt over several devices (or several groups on one device): every group is one mpst_sweep_batch on its own host
// thread - no collective, nothing shared between groups.  What scales on a node: the sharded sweep replicates its eigensolver
// (70 % of a bond) on every rank, independent fits do not.
int mpst_sweep_batch_multi(void* const* ctxs, int32_t K, const int32_t* group, mpst_sweep_stats* out) {
    if (!ctxs || K < 1 || K > 512) return fail(nullptr, MPST_ERR_INVALID, "mpst_sweep_batch_multi: 1..512 contexts");
    for (int k = 0; k < K; ++k)
        if (!ctxs[k]) return fail(nullptr, MPST_ERR_INVALID, "context %d is NULL", k);
    // groups in order of first appearance; default: one group per device
    std::vector<int> gid(K), keys;
    for (int k = 0; k < K; ++k) {
        const int key = group ? group[k] : ((Ctx*)ctxs[k])->device;
        int g = -1;
        for (size_t j = 0; j < keys.size(); ++j)
            if (keys[j] == key) g = (int)j;
        if (g < 0) {
            keys.push_back(key);
            g = (int)keys.size() - 1;
        }
        gid[k] = g;
    }
    const int G = (int)keys.size();
    std::vector<std::vector<void*>> members(G);
    std::vector<std::vector<int>> index(G);
    for (int k = 0; k < K; ++k) {
        members[gid[k]].push_back(ctxs[k]);
        index[gid[k]].push_back(k);
    }
    for (int g = 0; g < G; ++g) {
        if (members[g].size() > 64) return fail((Ctx*)ctxs[0], MPST_ERR_INVALID, "group %d holds %zu contexts (at most 64 per group)", keys[g], members[g].size());
        for (void* m : members[g])
            if (((Ctx*)m)->device != ((Ctx*)members[g][0])->device)
                return fail((Ctx*)ctxs[0], MPST_ERR_INVALID, "group %d mixes devices %d and %d: a group is one launch chain on one device", keys[g],
                            ((Ctx*)members[g][0])->device, ((Ctx*)m)->device);
    }
    // a context in two groups would be swept by two host threads at once, on one stream and one set of device scalars
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail((Ctx*)ctxs[0], MPST_ERR_INVALID, "context %d appears twice", k);
    std::vector<int> rc(G, 0);
    std::vector<std::vector<mpst_sweep_stats>> st(G);
    // every group's validation and graph capture here, one after the other; the threads only replay
    for (int g = 0; g < G; ++g) {
        st[g].resize(members[g].size());
        if ((rc[g] = sweep_batch_impl(members[g].data(), (int32_t)members[g].size(), st[g].data(), 1))) {
            Ctx* lead = (Ctx*)members[g][0];
            if (lead != (Ctx*)ctxs[0]) ((Ctx*)ctxs[0])->err = lead->err;
            return rc[g];
        }
    }
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g] { rc[g] = sweep_batch_impl(members[g].data(), (int32_t)members[g].size(), st[g].data(), 2); });
    for (auto& t : th) t.join();
    if (out)
        for (int g = 0; g < G; ++g)
            for (size_t j = 0; j < index[g].size(); ++j) out[index[g][j]] = st[g][j];
    for (int g = 0; g < G; ++g)
        if (rc[g]) {
            Ctx* lead = (Ctx*)members[g][0];
            if (lead != (Ctx*)ctxs[0]) ((Ctx*)ctxs[0])->err = lead->err;      // errors are reported on ctxs[0]
            return rc[g];
        }
    return 0;
}

int mpst_set_batch_hint(void* ctx, int32_t K) {
    Ctx* c = (Ctx*)ctx;
    if (!c || K < 1) return fail(c, MPST_ERR_INVALID, "batch hint must be >= 1");
    if (K != c->batch_hint) {
        c->batch_hint = K;
        c->ws_ready = false;
        c->epoch++;
    }
    return 0;
}

int mpst_get_loss_trace(void* ctx, double* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out) return MPST_ERR_INVALID;
    int rc = check_ready(c);
    if (rc) return rc;
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemcpy(out, c->loss_trace, (size_t)2 * (c->T - 1) * (c->opt.update_iters + 1) * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int mpst_bond_step(void* ctx, int32_t lid, int32_t going_left, mpst_bond_debug* dbg) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    if (lid < 0 || lid > c->T - 2) return fail(c, MPST_ERR_INVALID, "lid out of range");
    if (!c->caches_valid)
        return fail(c, MPST_ERR_INVALID, "the environment caches do not describe the current MPS / data set: call mpst_build_caches first");
    if (c->host_label_site != lid && c->host_label_site != lid + 1)
        return fail(c, MPST_ERR_INVALID, "bond (%d,%d) does not hold the label index (it is on site %d)", lid, lid + 1, c->host_label_site);
    View v = make_view(c, MPST_TRAIN);
    if ((rc = enqueue_reset_status(c))) return rc;
    if ((rc = enqueue_bond(c, v, lid, going_left ? 1 : 0))) return rc;
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    DevScalars sc;
    HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    if (sc.redo > 0) {
        // the tail launch of the four-launch chain left the bond alone (its verification failed): once more on the six-launch chain
        c->tail_redos++;
        c->chain4_hold = true;
        struct Hold { bool& h; ~Hold() { h = false; } } hold{c->chain4_hold};
        View v6 = make_view(c, MPST_TRAIN);
        if ((rc = enqueue_reset_status(c))) return rc;
        if ((rc = enqueue_bond(c, v6, lid, going_left ? 1 : 0))) return rc;
        HIPC(c, hipGetLastError());
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    }
    refresh_ss_counts(c);
    c->host_label_site = going_left ? lid : lid + 1;
    prof_collect(c);
    if (dbg) {
        std::vector<double> lam((size_t)std::max(sc.n_spec, 1));
        HIPC(c, hipMemcpy(lam.data(), c->lam, lam.size() * sizeof(double), hipMemcpyDeviceToHost));
        dbg->loss = sc.loss;
        dbg->grad_norm = sc.grad_norm;
        dbg->bt_norm = std::sqrt(sc.bt_norm2);
        dbg->chi_new = sc.n_keep;
        dbg->n_spectrum = sc.n_spec;
        dbg->eig_sweeps = sc.eig_sweeps;
        dbg->reserved = 0;
        for (int i = 0; i < sc.n_spec && i < MPST_MAX_SPECTRUM; ++i) dbg->spectrum[i] = std::sqrt(std::max(lam[i], 0.0)) * sc.inv_norm;
    }
    if (sc.status) {
        c->caches_valid = false;
        return fail(c, MPST_ERR_SVD, "bond-tensor decomposition failed at bond %d", lid);
    }
    return 0;
}

int mpst_eval(void* ctx, int which, double* mse, double* kld, double* acc, int64_t* conf) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    if (which != 0 && which != 1) return fail(c, MPST_ERR_INVALID, "which must be 0 or 1");
    if ((rc = enqueue_eval(c, which))) return rc;
    View v = make_view(c, which);
    if (c->typed) launch_teval_reduce(make_tview(c, which), c->yeval, c->out3, c->conf, c->pred, c->stream);
    else launch_eval_reduce(v, c->yeval, c->out3, c->conf, c->pred, c->stream);
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    double o[3];
    HIPC(c, hipMemcpy(o, c->out3, sizeof o, hipMemcpyDeviceToHost));
    // multi-GPU: sums over shards
    double tot[4] = {o[0], o[1], o[2], (double)v.N};
    std::vector<int64_t> cf((size_t)c->C * c->C);
    HIPC(c, hipMemcpy(cf.data(), c->conf, cf.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (multi(c)) {
        // sums over shards: the 4 scalars and the C x C counts (exact in fp64) travel as one message
        std::vector<double> msg(4 + cf.size());
        for (int i = 0; i < 4; ++i) msg[i] = tot[i];
        for (size_t i = 0; i < cf.size(); ++i) msg[4 + i] = (double)cf[i];
        double* dmsg = c->yeval;                       // scratch of at least N*C >= ... doubles; the message is small
        if ((int64_t)msg.size() > c->eval_N * c->C) return fail(c, MPST_ERR_INVALID, "evaluation scratch too small for the all-reduce message");
        HIPC(c, hipMemcpy(dmsg, msg.data(), msg.size() * sizeof(double), hipMemcpyHostToDevice));
        if ((rc = enqueue_allreduce(c, dmsg, (int64_t)msg.size(), -1))) return rc;
        HIPC(c, hipStreamSynchronize(c->stream));
        HIPC(c, hipMemcpy(msg.data(), dmsg, msg.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) tot[i] = msg[i];
        for (size_t i = 0; i < cf.size(); ++i) cf[i] = (int64_t)llround(msg[4 + i]);
    }
    if (mse) *mse = tot[0] / tot[3];
    if (kld) *kld = tot[1] / tot[3];
    if (acc) *acc = tot[2] / tot[3];
    if (conf) memcpy(conf, cf.data(), cf.size() * sizeof(int64_t));
    return 0;
}

int mpst_classify(void* ctx, int which, int32_t* pred, double* yhat) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    if (which != 0 && which != 1) return fail(c, MPST_ERR_INVALID, "which must be 0 or 1");
    if ((rc = enqueue_eval(c, which))) return rc;
    View v = make_view(c, which);
    if (c->typed) launch_teval_reduce(make_tview(c, which), c->yeval, c->out3, c->conf, c->pred, c->stream);
    else launch_eval_reduce(v, c->yeval, c->out3, c->conf, c->pred, c->stream);
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    if (pred) HIPC(c, hipMemcpy(pred, c->pred, (size_t)v.N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (yhat) {
        if (c->typed && c->zw == 1) {           // the typed kernels keep (re, im) pairs: a real context returns the real parts
            std::vector<double> tmp((size_t)v.N * c->C * 2);
            HIPC(c, hipMemcpy(tmp.data(), c->yeval, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < (size_t)v.N * c->C; ++i) yhat[i] = tmp[2 * i];
        } else {
            HIPC(c, hipMemcpy(yhat, c->yeval, (size_t)v.N * c->C * c->zw * sizeof(double), hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

// Are the grid states the Fourier basis (src/Encodings/bases.jl:23-42: cispi(f_s x) / sqrt(d), f = 0, 1, -1, 2, -2, ...) on a uniform
// grid?  Then the kernels evaluate the conditional densities and their cumulative sums in closed form instead of streaming the
// table (k_imp_left<..., TRIG>).  Decided from the tables the caller handed over, to 1e-12; MPST_IMP_NO_TRIG=1 keeps the table path.
static bool fourier_grid(const double* gx, const double* gp, int n, int d, double* x0, double* dxu) {
    if (getenv("MPST_IMP_NO_TRIG")) return false;
    const double a = gx[0], h = (gx[n - 1] - gx[0]) / (double)(n - 1);
    if (!(h > 0.0) || !((d - 1) * h < 1.0)) return false;
    const double tolx = 1e-12 * std::max(1.0, std::max(fabs(a), fabs(gx[n - 1])));
    const double inv = 1.0 / sqrt((double)d);
    for (int k = 0; k < n; ++k) {
        if (fabs(gx[k] - (a + k * h)) > tolx) return false;
        for (int s = 0; s < d; ++s) {
            const int f = (s + 1) / 2 * ((s & 1) ? 1 : -1);
            const double ang = M_PI * (double)f * gx[k];
            if (fabs(gp[2 * ((size_t)k * d + s)] - cos(ang) * inv) > 1e-12 || fabs(gp[2 * ((size_t)k * d + s) + 1] - sin(ang) * inv) > 1e-12) return false;
        }
    }
    *x0 = a;
    *dxu = h;
    return true;
}

// Are the grid states one of the Legendre bases (bases.jl:70-108: sqrt((2s+1)/2) P_s(x), optionally over sqrt(sqrt((2d+1)/2) d)) on a
// uniform grid inside [-1, 1]?  Then p(x) is a Legendre series of degree 2d - 2 and the kernels need no table of grid states
// (k_imp_left<..., TRIG>, real models).  On success `lin` receives the linearisation table A[l][s][s'] = cn^2 kappa_s kappa_s'
// a(s, s', l), P_s P_s' = sum_l a(s, s', l) P_l, by Gauss-Legendre quadrature (64 nodes: exact far beyond degree 3 x 30).
static bool legendre_grid(const double* gx, const double* gp, int n, int d, double* x0, double* dxu, std::vector<double>* lin) {
    if (getenv("MPST_IMP_NO_TRIG") || d < 1 || d > 16) return false;
    const double a = gx[0], h = (gx[n - 1] - gx[0]) / (double)(n - 1);
    if (!(h > 0.0) || a < -1.0 - 1e-12 || gx[n - 1] > 1.0 + 1e-12) return false;
    // the closed-form prefix sums are Euler-Maclaurin truncated after the h^3 (third derivative) term; the next one,
    // h^5 / 30240 * delta p^(5), grows like (2d - 2)^10 for a Legendre series of degree 2d - 2: on a coarse grid it moves the cumulative
    // trapezoid by whole grid steps (6e-4 relative at d = 16 with 101 points).  Only grids on which it stays below 1e-12 take the
    // closed form, the rest the table path.
    if (pow(h, 5.0) * pow(2.0 * d - 2.0, 10.0) / 1e8 >= 1e-12) return false;
    const double tolx = 1e-12;
    const double nrm = sqrt(sqrt((2 * d + 1) / 2.0) * d);
    double cn = 0.0;
    std::vector<double> P(d);
    for (int k = 0; k < n; ++k) {
        const double x = gx[k];
        if (fabs(x - (a + k * h)) > tolx) return false;
        P[0] = 1.0;
        if (d > 1) P[1] = x;
        for (int m = 1; m + 1 < d; ++m) P[m + 1] = ((2 * m + 1) * x * P[m] - m * P[m - 1]) / (m + 1);
        if (k == 0) {
            // the scale of the table: with or without the norm (state 0 is the constant sqrt(1/2) cn)
            const double c0 = gp[0] / sqrt(0.5);
            if (fabs(c0 - 1.0) < 1e-12) cn = 1.0;
            else if (fabs(c0 - 1.0 / nrm) < 1e-12) cn = 1.0 / nrm;
            else return false;
        }
        for (int s = 0; s < d; ++s)
            if (fabs(gp[(size_t)k * d + s] - cn * sqrt((2.0 * s + 1.0) / 2.0) * P[s]) > 1e-12) return false;
    }
    // Gauss-Legendre nodes and weights (Newton on P_64)
    const int NG = 64, L = 2 * d - 1;
    std::vector<double> xg(NG), wg(NG);
    for (int i = 0; i < NG; ++i) {
        double x = cos(M_PI * (i + 0.75) / (NG + 0.5)), dp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p0 = 1.0, p1 = x;
            for (int m = 1; m < NG; ++m) {
                const double p2 = ((2 * m + 1) * x * p1 - m * p0) / (m + 1);
                p0 = p1;
                p1 = p2;
            }
            dp = NG * (x * p1 - p0) / (x * x - 1.0);
            const double dxn = p1 / dp;
            x -= dxn;
            if (fabs(dxn) < 1e-16) break;
        }
        xg[i] = x;
        wg[i] = 2.0 / ((1.0 - x * x) * dp * dp);
    }
    lin->assign((size_t)L * d * d, 0.0);
    std::vector<double> Q(L);
    for (int i = 0; i < NG; ++i) {
        const double x = xg[i];
        Q[0] = 1.0;
        if (L > 1) Q[1] = x;
        for (int m = 1; m + 1 < L; ++m) Q[m + 1] = ((2 * m + 1) * x * Q[m] - m * Q[m - 1]) / (m + 1);
        for (int l = 0; l < L; ++l)
            for (int s = 0; s < d; ++s)
                for (int t = 0; t < d; ++t)
                    (*lin)[((size_t)l * d + s) * d + t] += wg[i] * 0.5 * (2 * l + 1) * Q[l] * Q[s] * Q[t];
    }
    for (int l = 0; l < L; ++l)
        for (int s = 0; s < d; ++s)
            for (int t = 0; t < d; ++t)
                (*lin)[((size_t)l * d + s) * d + t] *= cn * cn * sqrt((2.0 * s + 1.0) / 2.0) * sqrt((2.0 * t + 1.0) / 2.0);
    *x0 = a;
    *dxu = h;
    return true;
}

// shared tail of the two imputation entry points: option checks, scratch, launches, results
static int run_impute(Ctx* c, const ImpModel& m, const uint8_t* missing, const double* grid_x, const void* grid_phi, int32_t ngrid,
                      const mpst_impute_opts* o, const double* u, double* x_out, double* err_out, double* seconds) {
    if (!missing || !grid_x || !grid_phi || !x_out || !o || ngrid < 2) return fail(c, MPST_ERR_INVALID, "NULL argument or fewer than 2 grid values");
    const int method = o->method;
    if (method < MPST_IMPUTE_MEDIAN || method > MPST_IMPUTE_ITS_REJECT) return fail(c, MPST_ERR_INVALID, "unknown imputation method");
    if (o->order != MPST_IMPUTE_FORWARDS && o->order != MPST_IMPUTE_BACKWARDS) return fail(c, MPST_ERR_INVALID, "impute_order must be forwards (0) or backwards (1)");
    const bool sampling = method == MPST_IMPUTE_QUANTILE || method == MPST_IMPUTE_ITS_REJECT;
    if (sampling && !u) return fail(c, MPST_ERR_INVALID, "the sampling methods need the uniform numbers u[N][T][max_trials]");
    const int ntrial = method == MPST_IMPUTE_ITS_REJECT ? o->max_trials : 1;
    if (ntrial < 1) return fail(c, MPST_ERR_INVALID, "max_trials must be at least 1");
    if (method == MPST_IMPUTE_ITS_REJECT && !(o->rejection_threshold >= 0.0)) return fail(c, MPST_ERR_INVALID, "rejection_threshold must be non-negative");
    if (method == MPST_IMPUTE_MEAN) {
        const int mb = o->mean_basis;
        const bool real_ok = mb == MPST_BASIS_LEGENDRE || mb == MPST_BASIS_LEGENDRE_NO_NORM || mb == MPST_BASIS_UNIFORM;
        const bool cplx_ok = mb == MPST_BASIS_FOURIER || (mb == MPST_BASIS_STOUDENMIRE && m.d == 2) || (mb == MPST_BASIS_SAHAND && m.d % 2 == 0);
        if (!(m.is_complex ? cplx_ok : real_ok))
            return fail(c, MPST_ERR_UNSUPPORTED, "the mean method re-encodes on the device: Legendre / Uniform bases (real models), Fourier, "
                                                 "Stoudenmire (d = 2) or Sahand (even d) (complex models) only");
    }
    const int lim = impute_chi_limit(m.is_complex != 0, m.compute_f32 != 0);
    if (m.cap > lim || m.d > 16)
        return fail(c, MPST_ERR_UNSUPPORTED, "the imputation engine holds chi_max <= %d (this element type) and d <= 16 (got %d, %d)", lim, m.cap, m.d);
    hipError_t ea = impute_init_attrs(c->device);
    if (ea != hipSuccess) return fail(c, MPST_ERR_DEVICE, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));
    const int64_t N = m.N;
    const int T = m.T, d = m.d;
    const int zw = m.is_complex ? 2 : 1;
    const size_t esz = m.compute_f32 ? 4 : 8;
    int maxm = 0;
    for (int64_t i = 0; i < N; ++i) {
        int mm = 0;
        for (int j = 0; j < T; ++j) mm += missing[i * T + j] ? 1 : 0;
        maxm = std::max(maxm, mm);
    }
    std::vector<double> xo((size_t)N * T, 0.0), eo((size_t)N * T, 0.0);
    // Instances are dealt out to workgroups in the order of where their missing sites begin (in the direction of the sweep):
    // workgroups that are resident together then walk the chain in step and find the site tensor the first of them fetched
    // still in the L2 (a 262 KB tensor per site at configs[4], re-read by every instance).  Results do not depend on the order.
    std::vector<int32_t> order((size_t)N);
    {
        std::vector<int32_t> key((size_t)N, T);
        const bool backwards = o->order == MPST_IMPUTE_BACKWARDS;
        for (int64_t i = 0; i < N; ++i) {
            order[i] = (int32_t)i;
            for (int j = 0; j < T; ++j)
                if (missing[i * T + (backwards ? T - 1 - j : j)]) { key[i] = j; break; }
        }
        if (getenv("MPST_IMP_NO_ORDER") == nullptr)
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    }
    if (maxm > 0) {
        // instances are processed in chunks so that the per-instance scratch (environments of the missing sites, p_k and
        // its prefix sums) stays below half of the free device memory, at most 48 GB (MPST_IMPUTE_CHUNK_GB overrides): the
        // 27 GB of environments of configs[4] (8192 instances x 100 missing sites x 32 KB) are one chunk on a 288 GB device,
        // 512 workgroups of the batched sweep instead of seven launches of 82
        const int64_t welems = impute_work_elems(m.cap, m.is_complex != 0, m.compute_f32 != 0);
        const int64_t per_bytes = ((int64_t)maxm * m.cap * m.cap * zw + welems) * (int64_t)esz + 2ll * ngrid * (int64_t)sizeof(double);
        size_t free_b = 0, total_b = 0;
        HIPC(c, hipMemGetInfo(&free_b, &total_b));
        double budget = std::min(48.0 * (double)(1ull << 30), 0.5 * (double)free_b);
        if (const char* e = getenv("MPST_IMPUTE_CHUNK_GB")) budget = std::max(0.001, atof(e)) * (double)(1ull << 30);
        int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(N, (int64_t)(budget / (double)per_bytes)));
        if (chunk < N && chunk > 4096) chunk &= ~(int64_t)4095;      // whole rounds of 16-instance workgroups on 256 CUs
        uint8_t *dmiss = nullptr, *dR = nullptr, *dW = nullptr;
        int32_t* dord = nullptr;
        double *dgx = nullptr, *dgp = nullptr, *du = nullptr, *dp = nullptr, *dS = nullptr, *dx = nullptr, *de = nullptr, *dlin = nullptr;
        struct Temps {
            uint8_t **m, **r, **w; double **b, **cc, **dd, **e, **f, **g, **h; int32_t** o; double** l;
            ~Temps() { dfree(m); dfree(r); dfree(w); dfree(b); dfree(cc); dfree(dd); dfree(e); dfree(f); dfree(g); dfree(h); dfree(o); dfree(l); }
        } temps{&dmiss, &dR, &dW, &dgx, &dgp, &du, &dp, &dS, &dx, &de, &dord, &dlin};
        int rc;
        if ((rc = dalloc(c, &dmiss, N * T)) || (rc = dalloc(c, &dR, (int64_t)(chunk * maxm * m.cap * m.cap * zw * esz))) ||
            (rc = dalloc(c, &dgx, ngrid)) || (rc = dalloc(c, &dgp, (int64_t)ngrid * d * zw)) || (rc = dalloc(c, &dp, chunk * ngrid)) ||
            (rc = dalloc(c, &dS, chunk * ngrid)) || (rc = dalloc(c, &dx, N * T)) || (rc = dalloc(c, &de, N * T))) return rc;
        if (sampling && (rc = dalloc(c, &du, N * T * ntrial))) return rc;
        if ((rc = dalloc(c, &dord, N))) return rc;
        HIPC(c, hipMemcpy(dord, order.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice));
        if (welems && (rc = dalloc(c, &dW, (int64_t)(chunk * welems * esz)))) return rc;
        HIPC(c, hipMemcpy(dmiss, missing, (size_t)N * T, hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(dgx, grid_x, (size_t)ngrid * sizeof(double), hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(dgp, grid_phi, (size_t)ngrid * d * zw * sizeof(double), hipMemcpyHostToDevice));
        if (sampling) HIPC(c, hipMemcpy(du, u, (size_t)N * T * ntrial * sizeof(double), hipMemcpyHostToDevice));
        HIPC(c, hipMemset(dx, 0, (size_t)N * T * sizeof(double)));
        HIPC(c, hipMemset(de, 0, (size_t)N * T * sizeof(double)));
        HIPC(c, hipEventRecord(c->ev_start, c->stream));
        double gx0 = 0.0, gdx = 0.0;
        std::vector<double> lin;
        const int trig = m.is_complex ? (fourier_grid(grid_x, (const double*)grid_phi, ngrid, d, &gx0, &gdx) ? 1 : 0)
                                      : (legendre_grid(grid_x, (const double*)grid_phi, ngrid, d, &gx0, &gdx, &lin) ? 1 : 0);
        c->impute_trig = trig;
        if (!lin.empty()) {
            if ((rc = dalloc(c, &dlin, (int64_t)lin.size()))) return rc;
            HIPC(c, hipMemcpy(dlin, lin.data(), lin.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        const ImputeParams q{dmiss, dR, dW, dgx, dgp, du, dp, dS, dx, de, maxm, ngrid, method, o->get_err, o->order == MPST_IMPUTE_BACKWARDS ? 1 : 0,
                             ntrial, o->mean_basis, o->rejection_threshold, trig, gx0, gdx, dord, dlin};
        // one event between the two kernels of every chunk: the split of the pass into its environment and density halves
        // (mpst_get_impute_phases) costs nothing against kernels of tens of milliseconds
        struct Evs {
            std::vector<hipEvent_t> e;
            ~Evs() { for (auto x : e) (void)hipEventDestroy(x); }
        } evs;
        for (int64_t i0 = 0; i0 < N; i0 += chunk) {
            hipEvent_t mid = nullptr, end = nullptr;
            HIPC(c, hipEventCreate(&mid));
            evs.e.push_back(mid);
            HIPC(c, hipEventCreate(&end));
            evs.e.push_back(end);
            c->impute_batched = launch_impute(m, q, i0, std::min(chunk, N - i0), c->stream, mid);
            HIPC(c, hipEventRecord(end, c->stream));
        }
        HIPC(c, hipEventRecord(c->ev_stop, c->stream));
        HIPC(c, hipGetLastError());
        HIPC(c, hipEventSynchronize(c->ev_stop));
        float ms = 0.f;
        HIPC(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
        if (seconds) *seconds = 1e-3 * ms;
        c->impute_phase_s[0] = c->impute_phase_s[1] = 0.0;
        for (size_t k = 0; k < evs.e.size(); k += 2) {
            float a = 0.f, b = 0.f;
            HIPC(c, hipEventElapsedTime(&a, k == 0 ? c->ev_start : evs.e[k - 1], evs.e[k]));
            HIPC(c, hipEventElapsedTime(&b, evs.e[k], evs.e[k + 1]));
            c->impute_phase_s[0] += 1e-3 * a;
            c->impute_phase_s[1] += 1e-3 * b;
        }
        if (const char* e = getenv("MPST_IMB_DBG")) {
            if (atoi(e) & 8) {           // lab: phase clocks of the batched sweep (k_imp_leftb, workgroup 0 of the last chunk)
                double ph[6] = {0, 0, 0, 0, 0, 0};
                HIPC(c, hipMemcpy(ph, dp, sizeof(ph), hipMemcpyDeviceToHost));
                fprintf(stderr, "[imb] phase A %.0f us, barrier %.0f, B1 %.0f, B2 %.0f, barrier %.0f\n", ph[0] * 0.01, ph[1] * 0.01, ph[2] * 0.01,
                        ph[3] * 0.01, ph[4] * 0.01);
            }
        }
        HIPC(c, hipMemcpy(xo.data(), dx, xo.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPC(c, hipMemcpy(eo.data(), de, eo.size() * sizeof(double), hipMemcpyDeviceToHost));
    } else if (seconds) {
        *seconds = 0.0;
    }
    memcpy(x_out, xo.data(), xo.size() * sizeof(double));
    if (err_out) memcpy(err_out, eo.data(), eo.size() * sizeof(double));
    return 0;
}

int mpst_get_impute_phases(void* ctx, double* seconds_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !seconds_out) return MPST_ERR_INVALID;
    seconds_out[0] = c->impute_phase_s[0];
    seconds_out[1] = c->impute_phase_s[1];
    return 0;
}

int mpst_get_impute_info(void* ctx, int32_t* out, int32_t n) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out || n < 0) return MPST_ERR_INVALID;
    const int32_t full[2] = {c->impute_trig, c->impute_batched};
    for (int i = 0; i < n && i < 2; ++i) out[i] = full[i];
    return 0;
}

int mpst_impute(void* ctx, int which, const uint8_t* missing, const double* grid_x, const double* grid_phi, int32_t ngrid,
                const mpst_impute_opts* o, const double* u, double* x_out, double* err_out, double* seconds) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (which != MPST_TRAIN && which != MPST_TEST) return fail(c, MPST_ERR_INVALID, "which must be 0 or 1");
    if (!c->have_mps || !c->have_opt) return fail(c, MPST_ERR_INVALID, "mpst_set_options / mpst_set_mps must be called first");
    const DataSet& s = c->ds[which];
    if (s.N <= 0) return fail(c, MPST_ERR_INVALID, "data set %d is empty", which);
    HIPC(c, hipSetDevice(c->device));
    const View v = make_view(c, which);
    const ImpModel m{v.sites, v.site_stride, v.chi, v.label_site, v.phi, v.label, s.N, c->T, c->d, c->cap, c->zw == 2 ? 1 : 0,
                     (c->dtype == MPST_F32 || c->dtype == MPST_C64) ? 1 : 0};
    return run_impute(c, m, missing, grid_x, grid_phi, ngrid, o, u, x_out, err_out, seconds);
}

// Host arrays of a model in the boundary layouts (site: (s, l, r[, c]) column-major like mpst_set_mps; phi: [N][T][d]) to
// the engine's layouts and element type.
extern "C++" {
template <typename R>
static void pack_model(const mpst_impute_model* h, int cap, int64_t stride, bool cx, std::vector<R>& sites, std::vector<R>& phi) {
    const int T = h->T, d = h->d, C = h->C, zw = cx ? 2 : 1;
    sites.assign((size_t)stride * T * zw, R(0));
    for (int j = 0; j < T; ++j) {
        const int Dl = h->chi[j], Dr = h->chi[j + 1], Cj = (j == h->label_site) ? C : 1;
        const double* src = (const double*)h->site[j];
        R* dst = &sites[(size_t)j * stride * zw];
        for (int cc = 0; cc < Cj; ++cc)
            for (int r = 0; r < Dr; ++r)
                for (int l = 0; l < Dl; ++l)
                    for (int s = 0; s < d; ++s) {
                        const size_t to = (((size_t)cc * Dl + l) * d + s) * Dr + r, from = s + (size_t)d * (l + (size_t)Dl * (r + (size_t)Dr * cc));
                        for (int z = 0; z < zw; ++z) dst[to * zw + z] = (R)src[from * zw + z];
                    }
    }
    const int64_t N = h->N;
    phi.resize((size_t)N * T * d * zw);
    const double* ps = (const double*)h->phi;
    for (int64_t i = 0; i < N; ++i)
        for (int j = 0; j < T; ++j)
            for (int s = 0; s < d * zw; ++s) phi[((size_t)j * N + i) * d * zw + s] = (R)ps[((size_t)i * T + j) * d * zw + s];
}
}  // extern "C++"

int mpst_impute_model_run(void* ctx, const mpst_impute_model* h, const uint8_t* missing, const double* grid_x, const void* grid_phi,
                          int32_t ngrid, const mpst_impute_opts* o, const double* u, double* x_out, double* err_out, double* seconds) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (!h || !h->site || !h->chi || !h->phi || !h->label_idx) return fail(c, MPST_ERR_INVALID, "NULL argument");
    if (h->N <= 0 || h->T < 1 || h->d < 1 || h->C < 1) return fail(c, MPST_ERR_INVALID, "empty model or data");
    if (h->dtype != MPST_DTYPE_F64 && h->dtype != MPST_DTYPE_C64) return fail(c, MPST_ERR_INVALID, "dtype must be MPST_DTYPE_F64 or MPST_DTYPE_C64");
    if (h->compute != MPST_COMPUTE_F64 && h->compute != MPST_COMPUTE_F32) return fail(c, MPST_ERR_INVALID, "compute must be MPST_COMPUTE_F64 or MPST_COMPUTE_F32");
    if (h->label_site < 0 || h->label_site >= h->T) return fail(c, MPST_ERR_INVALID, "label_site out of range");
    if (h->chi[0] != 1 || h->chi[h->T] != 1) return fail(c, MPST_ERR_INVALID, "chi[0] and chi[T] must be 1");
    int cap = 1;
    for (int j = 0; j <= h->T; ++j) {
        if (h->chi[j] < 1) return fail(c, MPST_ERR_INVALID, "chi[%d] < 1", j);
        cap = std::max(cap, (int)h->chi[j]);
    }
    for (int j = 0; j < h->T; ++j)
        if (!h->site[j]) return fail(c, MPST_ERR_INVALID, "site[%d] is NULL", j);
    for (int64_t i = 0; i < h->N; ++i)
        if (h->label_idx[i] < 0 || h->label_idx[i] >= h->C) return fail(c, MPST_ERR_INVALID, "label_idx[%lld] out of range", (long long)i);
    const bool cx = h->dtype == MPST_DTYPE_C64, f32 = h->compute == MPST_COMPUTE_F32;
    HIPC(c, hipSetDevice(c->device));
    const int64_t stride = (int64_t)h->C * cap * h->d * cap;
    const size_t esz = (f32 ? 4 : 8) * (cx ? 2 : 1);
    uint8_t *dsites = nullptr, *dphi = nullptr;
    int32_t *dchi = nullptr, *dls = nullptr, *dlab = nullptr;
    struct Temps {
        uint8_t **a, **b; int32_t **x, **y, **z;
        ~Temps() { dfree(a); dfree(b); dfree(x); dfree(y); dfree(z); }
    } temps{&dsites, &dphi, &dchi, &dls, &dlab};
    int rc;
    if ((rc = dalloc(c, &dsites, (int64_t)(stride * h->T * esz))) || (rc = dalloc(c, &dphi, (int64_t)(h->N * h->T * h->d * esz))) ||
        (rc = dalloc(c, &dchi, h->T + 1)) || (rc = dalloc(c, &dls, 1)) || (rc = dalloc(c, &dlab, h->N))) return rc;
    if (f32) {
        std::vector<float> hs, hp;
        pack_model<float>(h, cap, stride, cx, hs, hp);
        HIPC(c, hipMemcpy(dsites, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(dphi, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<double> hs, hp;
        pack_model<double>(h, cap, stride, cx, hs, hp);
        HIPC(c, hipMemcpy(dsites, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(dphi, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPC(c, hipMemcpy(dchi, h->chi, (size_t)(h->T + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(dls, &h->label_site, sizeof(int32_t), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(dlab, h->label_idx, (size_t)h->N * sizeof(int32_t), hipMemcpyHostToDevice));
    const ImpModel m{dsites, stride, dchi, dls, dphi, dlab, h->N, h->T, h->d, cap, cx ? 1 : 0, f32 ? 1 : 0};
    return run_impute(c, m, missing, grid_x, grid_phi, ngrid, o, u, x_out, err_out, seconds);
}

int mpst_normalize(void* ctx) {
    Ctx* c = (Ctx*)ctx;
    int rc = check_ready(c);
    if (rc) return rc;
    View v = make_view(c, MPST_TRAIN);
    if (c->typed) {
        TView t = make_tview(c, MPST_TRAIN);
        launch_tnorm2(t, c->norm2, c->tnorm_scratch, c->stream);
        launch_tscale_sites(t, c->norm2, c->stream);
    } else {
        launch_norm2(v, c->norm2, c->norm_scratch, c->stream);
        launch_scale_sites(v, c->norm2, c->stream);
    }
    c->ynext_lid = -1;
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
}

int mpst_set_profile(void* ctx, uint32_t kernel_mask) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    c->prof_mask = kernel_mask;
    c->epoch++;
    for (int i = 0; i < 16; ++i) { c->prof_us[i] = 0; c->prof_cnt[i] = 0; }
    return 0;
}

int mpst_get_profile(void* ctx, double* total_us, int64_t* count) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    for (int i = 0; i < 16; ++i) {
        if (total_us) total_us[i] = c->prof_us[i];
        if (count) count[i] = c->prof_cnt[i];
    }
    return 0;
}

int mpst_get_info(void* ctx, int32_t* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !out) return MPST_ERR_INVALID;
    int rc = check_ready(c);
    if (rc) return rc;
    const int pk = c->opt.loss == MPST_LOSS_MSE ? 1 : 0;
    out[0] = c->fused ? 1 : 0;
    out[1] = c->big ? 1 : 0;
    out[2] = c->ds[MPST_TRAIN].nparts[pk];
    out[3] = c->ds[MPST_TRAIN].nchunks;
    out[4] = c->cap;
    out[5] = c->nranks;
    out[6] = (!multi(c) && !c->big && c->prof_mask == 0 && getenv("MPST_NO_GRAPH") == nullptr) ? 1 : 0;
    out[7] = (int32_t)std::min<int64_t>(c->big_fallbacks, 1 << 30);
    out[8] = blocked_eig_coop_aborts(c->blk);      // bonds the persistent tridiagonalisation handed back to the launch-per-step path
    out[9] = blocked_eig_xcd_misplaced(c->blk);    // bonds whose XCD-local attempt found its workgroups on several XCDs (redone across the XCDs)
    out[10] = c->b2 ? 1 : 0;                        // fused chain with the sliced bond GEMMs (k_yhat_s + k_grad_s)
    out[11] = c->b2 ? c->b2_ksplit : 0;             // shares per gradient block of k_grad_s
    out[12] = (!c->big && eig_merged()) ? 1 : 0;    // tridiagonalisation + eigenvectors in one launch (k_eig_trivec)
    out[13] = c->big_redos;                          // large-bond sweeps redone bond by bond after a failed verdict
    out[14] = (c->big_opt && !multi(c)) ? 1 : 0;                   // large bonds: the eigensolver's verdict is read once per sweep
    out[15] = c->typed ? 1 + c->dtype : 0;        // element-typed kernels in use: 1 + dtype
    return 0;
}

int mpst_get_info_n(void* ctx, int32_t* out, int32_t n) {
    int32_t full[20];
    if (!out || n < 0) return MPST_ERR_INVALID;
    int rc = mpst_get_info(ctx, full);
    if (rc) return rc;
    {
        Ctx* c4 = (Ctx*)ctx;
        View v4 = make_view(c4, MPST_TRAIN);
        // [18] the bonds of a sweep run the four-launch chain (k_bond_tail); [19] sweeps / bond steps whose tail was redone on the six-launch chain
        full[18] = (c4->chain4_ok && c4->b2 && !multi(c4) && c4->opt.update_iters == 1 && !c4->opt.track_cost && eig_merged() && bond_tail_supported(v4)) ? 1 : 0;
        full[19] = c4->tail_redos;
    }
    // bonds the subspace eigensolver attempted / whose result was accepted (the rest went to the exact solver), as of the last sweep,
    // batch or bond step that returned: cached at their synchronisation point, the query itself never waits for the stream
    full[16] = ((Ctx*)ctx)->ss_counts[0];
    full[17] = ((Ctx*)ctx)->ss_counts[1];
    for (int i = 0; i < n && i < 20; ++i) out[i] = full[i];
    return 0;
}

int mpst_get_eig_phases(void* ctx, double* us) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->sc || !us) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    DevScalars sc;
    HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    const unsigned long long* t = sc.eig_stamps;              // 100 MHz ticks
    us[0] = 0.01 * (double)(t[1] - t[0]);                     // k_eig_tri: tridiagonalisation
    us[1] = 0.01 * (double)(t[3] - t[2]);                     // k_eig_vec block 0: staging + bisection
    us[2] = 0.01 * (double)(t[4] - t[3]);                     //                    twisted factorisation
    us[3] = 0.01 * (double)(t[5] - t[4]);                     //                    back-transformation
    us[4] = 0.01 * (double)(t[9] - t[8]);                     // k_eig_fin: truncation, verification, Loewdin
    us[5] = (double)(t[7] - t[6]);                            // shader cycles spent in the tridiagonalisation
    return 0;
}

// phase stamps of the last stamped k_bond_tail launch (us since workgroup 0 started; -1: not taken): us[0..15] workgroup 0 (which also
// hosts a job of the next bond's tensor when the sweep goes on), us[16..31] the last workgroup (no role) (see include/mpstime_hip.h)
int mpst_get_tail_phases(void* ctx, double* us) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->sc || !us) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    DevScalars sc;
    HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    const unsigned long long* t = sc.eig_stamps;
    const double t0 = (double)t[16];
    // workgroup 0: slots 16..31; the last workgroup: 32..47
    for (int i = 0; i < 48; ++i) us[i] = -1.0;
    for (int i = 0; i < 16; ++i) if (t[16 + i] && t[16]) us[i] = 0.01 * ((double)t[16 + i] - t0);
    for (int i = 0; i < 16; ++i) if (t[32 + i] && t[16]) us[16 + i] = 0.01 * ((double)t[32 + i] - t0);
    // slots a launch did not reach keep the stamps of an earlier one: nothing after the first stamp that runs backwards counts
    for (int g = 0; g < 2; ++g)
        for (int i = 1; i < 16; ++i)
            if (us[16 * g + i] < us[16 * g + i - 1] || us[16 * g + i - 1] == -1.0) us[16 * g + i] = -1.0;
    for (int i = 0; i < 4; ++i) us[48 + i] = (double)t[56 + i];        // bonds by |Z^T Z - I| of their candidates: < 1e-13, < 1e-8, < 3e-5, above
    // every workgroup of the stamped launch left (start, end) in the gradient workspace: earliest start, latest end, latest start
    us[52] = us[53] = us[54] = 0.0;
    {
        const size_t ng = (size_t)std::min<unsigned long long>(t[60], 2048ull);
        if (ng > 0 && c->tail_span) {
            std::vector<unsigned long long> sp(2 * ng);
            HIPC(c, hipMemcpy(sp.data(), c->tail_span, sp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long s0 = ~0ull, s1 = 0ull, e1 = 0ull;
            for (size_t i = 0; i < ng; ++i) {
                s0 = std::min(s0, sp[2 * i]);
                s1 = std::max(s1, sp[2 * i]);
                e1 = std::max(e1, sp[2 * i + 1]);
            }
            us[52] = 0.01 * ((double)s0 - t0);
            us[53] = 0.01 * ((double)e1 - t0);
            us[54] = 0.01 * ((double)s1 - t0);
        }
    }
    return 0;
}

#ifdef MPST_B2_DEBUG
// bring-up builds only (scratch/build_dbg.sh): stamps of the sliced bond kernels, 8192 x 8 slots
int mpst_debug_b2(void* ctx, unsigned long long* out) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->b2_dbg || !out) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemcpy(out, c->b2_dbg, 8192 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

#endif

#ifdef MPST_TRI_DEBUG
// bring-up builds only (-DMPST_TRI_DEBUG): raw stamp slots of the eigensolver kernels
int mpst_debug_stamps(void* ctx, unsigned long long* out64) {
    Ctx* c = (Ctx*)ctx;
    if (!c || !c->sc || !out64) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    DevScalars sc;
    HIPC(c, hipMemcpy(&sc, c->sc, sizeof sc, hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; ++i) out64[i] = sc.eig_stamps[i];
    return 0;
}
#endif

int mpst_selftest_mfma(void* ctx, const double* A, const double* B, int32_t K, double* C_out) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    struct Temps { double **a, **b, **cc; ~Temps() { dfree(a); dfree(b); dfree(cc); } } temps{&dA, &dB, &dC};
    int rc;
    if ((rc = dalloc(c, &dA, 16 * K)) || (rc = dalloc(c, &dB, 16 * K)) || (rc = dalloc(c, &dC, 256))) return rc;
    HIPC(c, hipMemcpy(dA, A, (size_t)16 * K * sizeof(double), hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(dB, B, (size_t)16 * K * sizeof(double), hipMemcpyHostToDevice));
    launch_selftest_mfma(dA, dB, K, dC, c->stream);
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpy(C_out, dC, 256 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int mpst_selftest_eig(void* ctx, const double* G, int32_t n, int32_t alg, double* lambda_out, double* E_out, int32_t* sweeps) {
    Ctx* c = (Ctx*)ctx;
    if (!c) return MPST_ERR_INVALID;
    if (n < 1 || n > DIM_LIMIT) return fail(c, MPST_ERR_INVALID, "n must be in 1..%d", DIM_LIMIT);
    HIPC(c, hipSetDevice(c->device));
    double *dG = nullptr, *dl = nullptr, *dE = nullptr, *dws = nullptr;
    int32_t* ds = nullptr;
    struct Temps { double **a, **b, **cc, **d; int32_t** e; ~Temps() { dfree(a); dfree(b); dfree(cc); dfree(d); dfree(e); } }
        temps{&dG, &dl, &dE, &dws, &ds};
    hipError_t ea = eig_init_attrs(c->device);
    if (ea != hipSuccess) return fail(c, MPST_ERR_DEVICE, "hipFuncSetAttribute failed: %s", hipGetErrorString(ea));
    int rc;
    if ((rc = dalloc(c, &dG, n * n)) || (rc = dalloc(c, &dl, n)) || (rc = dalloc(c, &dE, n * n)) || (rc = dalloc(c, &ds, 1)) ||
        (rc = dalloc(c, &dws, (int64_t)eig_workspace_doubles()))) return rc;
    HIPC(c, hipMemcpy(dG, G, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    HIPC(c, hipMemset(dws, 0, eig_workspace_doubles() * sizeof(double)));
    if (n > MAX_DIM) {
        // alg 0: the hand-written blocked solver, library only if its verification asks for it; alg 2: the library alone
        std::string e;
        int need_lib = 1;
        View vraw{};
        vraw.zw = (alg & 4) ? 2 : 0;          // bit 2: G is the real embedding of a Hermitian matrix (one vector per eigenvalue pair)
        alg &= 3;
        if (alg != 2) {
            BlockedEig* bl = nullptr;
            if ((rc = blocked_eig_create(&bl, n, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
            HIPC(c, hipMemsetAsync(dl, 0, (size_t)n * sizeof(double), c->stream));
            HIPC(c, hipMemsetAsync(dE, 0, (size_t)n * n * sizeof(double), c->stream));
            need_lib = launch_eig_blocked(vraw, 0, 0, dG, n, dl, dE, ds, bl, c->stream);
            blocked_eig_destroy(bl);
            if (need_lib < 0) return fail(c, MPST_ERR_DEVICE, "blocked eigensolver failed");
        }
        if (need_lib && alg != 3) {
            BigEig* be = nullptr;
            if ((rc = big_eig_create(&be, n, c->stream, &e))) return fail(c, rc, "large-bond eigensolver: %s", e.c_str());
            rc = launch_eig_big_raw(dG, n, dl, dE, ds, be, c->stream);
            (void)hipStreamSynchronize(c->stream);
            big_eig_destroy(be);
            if (rc) return fail(c, rc, "the large-bond Jacobi solver could not be launched");
        }
    } else {
        launch_eig_raw(dG, n, alg, dl, dE, ds, dws, c->stream);
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpy(lambda_out, dl, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIPC(c, hipMemcpy(E_out, dE, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    if (sweeps) HIPC(c, hipMemcpy(sweeps, ds, sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
