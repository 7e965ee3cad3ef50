/*
 * mps_oracle.c - plain-C restatement of MPSTime.jl's training sweep (fp64, real).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the "cpu_baseline" (kind "port")
 * leg of bench.py.  Never linked into or called by the shipped library.
 * PARITY of the sweep itself UNPINNED against the Julia reference (it cannot run here); pinned against
 * oracle/ref_numpy.py (tests/test_oracle.py), which follows the same source lines and whose encoding /
 * contraction / container conventions ARE pinned on reference-produced tensors (see its header).
 *
 * It keeps the reference's loop structure so that it can stand in for "the reference CPU
 * path" when timed: one series at a time, the fused
 * phi-tilde / yhat / gradient loop with the one-sample-late divide
 * (src/Training/loss_functions.jl:248-262,322-379), a per-series cache update
 * (src/Training/RealRealHighDimension.jl:107-144), two full cache rebuilds per sweep
 * (:770,:804) and a dense LAPACK gesdd + the NDTensors truncation rule per bond (:146-203).
 * gesdd is the same LAPACK routine Julia's DivideAndConquer calls; it is passed in as a
 * function pointer (SciPy's bundled OpenBLAS/LAPACK, scipy.linalg.cython_lapack.dgesdd).
 *
 * Layouts: site tensor j = C-order array (Dl, d, Dr[, C]) in a slot of `slot` doubles;
 * phi[N][T][d]; LE/RE [T][N][cap].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*dgesdd_fn)(char* jobz, int* m, int* n, double* a, int* lda, double* s, double* u, int* ldu,
                          double* vt, int* ldvt, double* work, int* lwork, int* iwork, int* info);

typedef struct {
    int T, d, C, N, chi_max, cap;
    int update_iters, loss, opt, rescale_before, rescale_after, train_sep, rebuild_caches;
    double eta, cutoff;
} orc_opts;

typedef struct {
    const orc_opts* o;
    const double* phi;
    const int* label;
    const long* counts;
    double* sites;
    long slot;
    int* chi;
    int* label_site;
    double *LE, *RE;
    dgesdd_fn gesdd;
    /* scratch */
    double *bt, *grad, *kprev, *xa, *xb, *M, *U, *S, *Vt, *work;
    int* iwork;
    int lwork;
} orc_state;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* kron_conj2 (loss_functions.jl:193-200): out[j + l2*i] = x1[i]*x2[j] (real) */
static inline void kron2(const double* x1, int l1, const double* x2, int l2, double* out) {
    for (int i = 0; i < l1; ++i)
        for (int j = 0; j < l2; ++j) out[j + l2 * i] = x1[i] * x2[j];
}

/* kron_scaleadd_KLD! 4-vector form (loss_functions.jl:248-262): the fused inner loop */
static inline double kron_scaleadd(double* restrict k, double* restrict kprev, const double* restrict bt,
                                   double scale_mul, const double* restrict xa, int la, const double* restrict xb,
                                   int lb) {
    double yhat = 0.0;
    for (int i = 0; i < la; ++i) {
        const double xai = xa[i];
        double* kk = k + (long)lb * i;
        double* kp = kprev + (long)lb * i;
        const double* b = bt + (long)lb * i;
        for (int j = 0; j < lb; ++j) {
            const double ph = xai * xb[j];
            yhat += b[j] * ph;
            kk[j] += kp[j] * scale_mul;
            kp[j] = ph;
        }
    }
    return yhat;
}

static const double ONE = 1.0;

/* phi-tilde factors of one series (site-case dispatch of yhat_phitilde_KLD!!, :279-295):
 * xa = RE (x) ps[rid] (ps fastest), xb = LE (x) ps[lid] (ps fastest); missing environment = [1]. */
static inline void factors(const orc_state* st, int lid, long i, double* xa, int* la, double* xb, int* lb) {
    const orc_opts* o = st->o;
    const int rid = lid + 1, d = o->d;
    const int Dl = st->chi[lid], Dr = st->chi[rid + 1];
    const double* psl = st->phi + ((long)i * o->T + lid) * d;
    const double* psr = st->phi + ((long)i * o->T + rid) * d;
    const double* le = lid > 0 ? st->LE + ((long)(lid - 1) * o->N + i) * o->cap : &ONE;
    const double* re = rid < o->T - 1 ? st->RE + ((long)(rid + 1) * o->N + i) * o->cap : &ONE;
    kron2(re, Dr, psr, d, xa);
    kron2(le, Dl, psl, d, xb);
    *la = Dr * d;
    *lb = Dl * d;
}

/* Loss_Grad_KLD (:322-379 / :383-432) and Loss_Grad_MSE (:561-619) */
static double loss_grad(orc_state* st, int lid, double* grad) {
    const orc_opts* o = st->o;
    const int Dl = st->chi[lid], Dr = st->chi[lid + 2], d = o->d;
    const long L = (long)Dl * d * d * Dr;
    double losses = 0.0;
    memset(grad, 0, sizeof(double) * L * o->C);
    long i_prev = 0;
    double yprev = 0.0;
    for (int c = 0; c < o->C; ++c) {
        const long cn = st->counts[c];
        double yhat = 1.0;
        memset(st->kprev, 0, sizeof(double) * L);
        double* k = grad + L * c;
        const double* btc = st->bt + L * c;
        double loss = 0.0;
        int la, lb;
        if (o->loss == 0) {
            for (long i = i_prev; i < i_prev + cn; ++i) {
                factors(st, lid, i, st->xa, &la, st->xb, &lb);
                yhat = kron_scaleadd(k, st->kprev, btc, 1.0 / yhat, st->xa, la, st->xb, lb);
                loss += -log(yhat * yhat);
            }
            const double div = o->train_sep ? (double)cn : (double)o->N;
            if (o->train_sep) losses += loss / cn; else losses += loss;
            for (long e = 0; e < L; ++e) k[e] = -(k[e] + st->kprev[e] / yhat) / div;
        } else {
            for (long i = 0; i < o->N; ++i) {
                const double mask = (i >= i_prev && i < i_prev + cn) ? 1.0 : 0.0;
                factors(st, lid, i, st->xa, &la, st->xb, &lb);
                yhat = kron_scaleadd(k, st->kprev, btc, yhat - yprev, st->xa, la, st->xb, lb);
                loss += 0.5 * (yhat - mask) * (yhat - mask);
                yprev = mask;
            }
            losses += loss;
            for (long e = 0; e < L; ++e) k[e] = (k[e] + st->kprev[e] * (yhat - yprev)) / (double)o->N;
        }
        i_prev += cn;
    }
    if (o->loss == 1 || !o->train_sep) losses /= (double)o->N;
    return losses;
}

static double fro(const double* x, long n) {
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += x[i] * x[i];
    return sqrt(s);
}

/* NDTensors truncate! (relative cutoff) */
static int truncate_spectrum(const double* S, int n, int maxdim, double cutoff) {
    if (n == 1) return 1;
    double truncerr = 0.0, scale = 0.0;
    for (int i = 0; i < n; ++i) scale += S[i] * S[i];
    if (scale == 0.0) scale = 1.0;
    while (n > maxdim) { truncerr += S[n - 1] * S[n - 1]; --n; }
    while (n > 1 && truncerr + S[n - 1] * S[n - 1] <= cutoff * scale) { truncerr += S[n - 1] * S[n - 1]; --n; }
    return n;
}

/* per-series cache update (update_caches!, :107-144 / construct_caches :72-77,:90-98) */
static void env_left(const orc_state* st, int j, const double* W /* (Dl,d,Dr) */, int Dl, int Dr) {
    const orc_opts* o = st->o;
    for (long i = 0; i < o->N; ++i) {
        const double* ps = st->phi + ((long)i * o->T + j) * o->d;
        const double* le = j > 0 ? st->LE + ((long)(j - 1) * o->N + i) * o->cap : &ONE;
        double* out = st->LE + ((long)j * o->N + i) * o->cap;
        for (int k = 0; k < Dr; ++k) out[k] = 0.0;
        for (int a = 0; a < Dl; ++a)
            for (int s = 0; s < o->d; ++s) {
                const double f = le[a] * ps[s];
                const double* w = W + ((long)a * o->d + s) * Dr;
                for (int k = 0; k < Dr; ++k) out[k] += f * w[k];
            }
    }
}
static void env_right(const orc_state* st, int j, const double* W /* (Dl,d,Dr) */, int Dl, int Dr) {
    const orc_opts* o = st->o;
    for (long i = 0; i < o->N; ++i) {
        const double* ps = st->phi + ((long)i * o->T + j) * o->d;
        const double* re = j < o->T - 1 ? st->RE + ((long)(j + 1) * o->N + i) * o->cap : &ONE;
        double* out = st->RE + ((long)j * o->N + i) * o->cap;
        for (int k = 0; k < Dl; ++k) {
            double acc = 0.0;
            for (int s = 0; s < o->d; ++s) {
                const double* w = W + ((long)k * o->d + s) * Dr;
                double t = 0.0;
                for (int b = 0; b < Dr; ++b) t += w[b] * re[b];
                acc += ps[s] * t;
            }
            out[k] = acc;
        }
    }
}

static void construct_caches(orc_state* st, int going_left) {
    const orc_opts* o = st->o;
    if (going_left) {
        for (int j = 0; j <= o->T - 2; ++j) env_left(st, j, st->sites + st->slot * j, st->chi[j], st->chi[j + 1]);
    } else {
        for (int j = o->T - 1; j >= 1; --j) env_right(st, j, st->sites + st->slot * j, st->chi[j], st->chi[j + 1]);
    }
}

/* One bond: flatten_bt, apply_update, decomposeBT, update_caches! (:733-762 / :777-801).
 * dbg (may be NULL): [loss, grad_norm, bt_norm, chi_new, nspec, S...]. Returns 0 or LAPACK info. */
static int bond_step(orc_state* st, int lid, int going_left, double* dbg) {
    const orc_opts* o = st->o;
    const int rid = lid + 1, d = o->d, C = o->C;
    const int Dl = st->chi[lid], Dm = st->chi[rid], Dr = st->chi[rid + 1];
    const long L = (long)Dl * d * d * Dr;
    double* Wl = st->sites + st->slot * lid;
    double* Wr = st->sites + st->slot * rid;
    const int lab_left = (*st->label_site == lid);
    /* flatten_bt :221-238 - flat index s_l + d*(a + Dl*(s_r + d*b)) */
    for (int c = 0; c < C; ++c)
        for (int b = 0; b < Dr; ++b)
            for (int sr = 0; sr < d; ++sr)
                for (int a = 0; a < Dl; ++a)
                    for (int sl = 0; sl < d; ++sl) {
                        double acc = 0.0;
                        for (int m = 0; m < Dm; ++m) {
                            const double wl = lab_left ? Wl[(((long)a * d + sl) * Dm + m) * C + c] : Wl[((long)a * d + sl) * Dm + m];
                            const double wr = lab_left ? Wr[((long)m * d + sr) * Dr + b] : Wr[(((long)m * d + sr) * Dr + b) * C + c];
                            acc += wl * wr;
                        }
                        st->bt[L * c + sl + d * (a + (long)Dl * (sr + d * (long)b))] = acc;
                    }
    /* apply_update :88-188 */
    if (o->rescale_before) {
        const double nb = fro(st->bt, L * C);
        for (long e = 0; e < L * C; ++e) st->bt[e] /= nb;
    }
    for (int it = 0; it < o->update_iters; ++it) {
        const double loss = loss_grad(st, lid, st->grad);
        const double gn = fro(st->grad, L * C);
        if (dbg && it == 0) { dbg[0] = loss; dbg[1] = gn; }
        const double step = o->opt == 0 ? o->eta / gn : o->eta;
        for (long e = 0; e < L * C; ++e) st->bt[e] -= step * st->grad[e];
    }
    const double nb = fro(st->bt, L * C);
    if (dbg) dbg[2] = nb;
    if (o->rescale_after)
        for (long e = 0; e < L * C; ++e) st->bt[e] /= nb;
    /* decomposeBT :146-203 - column-major M for LAPACK */
    int m, n;
    if (going_left) {   /* rows (a, c, s_l) | cols (s_r, b) */
        m = Dl * C * d; n = d * Dr;
        for (int b = 0; b < Dr; ++b) for (int sr = 0; sr < d; ++sr) for (int a = 0; a < Dl; ++a)
            for (int c = 0; c < C; ++c) for (int sl = 0; sl < d; ++sl)
                st->M[(long)((a * C + c) * d + sl) + (long)m * (sr * Dr + b)] = st->bt[L * c + sl + d * (a + (long)Dl * (sr + d * (long)b))];
    } else {            /* rows (b, c, s_r) | cols (s_l, a) */
        m = Dr * C * d; n = d * Dl;
        for (int b = 0; b < Dr; ++b) for (int sr = 0; sr < d; ++sr) for (int a = 0; a < Dl; ++a)
            for (int c = 0; c < C; ++c) for (int sl = 0; sl < d; ++sl)
                st->M[(long)((b * C + c) * d + sr) + (long)m * (sl * Dl + a)] = st->bt[L * c + sl + d * (a + (long)Dl * (sr + d * (long)b))];
    }
    const int mn = m < n ? m : n;
    int info = 0, ldu = m, ldvt = mn, lda = m;
    char jobz = 'S';
    st->gesdd(&jobz, &m, &n, st->M, &lda, st->S, st->U, &ldu, st->Vt, &ldvt, st->work, &st->lwork, st->iwork, &info);
    if (info != 0) return info;
    const int nk = truncate_spectrum(st->S, mn, o->chi_max, o->cutoff);
    if (dbg) { dbg[3] = nk; dbg[4] = mn; for (int i = 0; i < mn; ++i) dbg[5 + i] = st->S[i]; }
    if (going_left) {
        /* left = U*S -> (a, s, k, c);  right = V -> (k, s, b) */
        for (int a = 0; a < Dl; ++a) for (int sl = 0; sl < d; ++sl) for (int k = 0; k < nk; ++k) for (int c = 0; c < C; ++c)
            Wl[(((long)a * d + sl) * nk + k) * C + c] = st->U[(long)((a * C + c) * d + sl) + (long)ldu * k] * st->S[k];
        for (int k = 0; k < nk; ++k) for (int sr = 0; sr < d; ++sr) for (int b = 0; b < Dr; ++b)
            Wr[((long)k * d + sr) * Dr + b] = st->Vt[k + (long)ldvt * (sr * Dr + b)];
        st->chi[rid] = nk;
        *st->label_site = lid;
        env_right(st, rid, Wr, nk, Dr);                      /* update_caches! :124-131 */
    } else {
        /* right = U_svd*S -> (k, s, b, c);  left = V -> (a, s, k) */
        for (int k = 0; k < nk; ++k) for (int sr = 0; sr < d; ++sr) for (int b = 0; b < Dr; ++b) for (int c = 0; c < C; ++c)
            Wr[(((long)k * d + sr) * Dr + b) * C + c] = st->U[(long)((b * C + c) * d + sr) + (long)ldu * k] * st->S[k];
        for (int a = 0; a < Dl; ++a) for (int sl = 0; sl < d; ++sl) for (int k = 0; k < nk; ++k)
            Wl[((long)a * d + sl) * nk + k] = st->Vt[k + (long)ldvt * (sl * Dl + a)];
        st->chi[rid] = nk;
        *st->label_site = rid;
        env_left(st, lid, Wl, Dl, nk);                       /* :133-141 */
    }
    return 0;
}

/* the truncation rule alone, for tests */
int orc_truncate(const double* S, int n, int maxdim, double cutoff) { return truncate_spectrum(S, n, maxdim, cutoff); }

/* Exported entry point.  mode 0: bond updates of one full sweep (:727-808), numbered 0..2(T-1)-1 in
 * sweep order (backward half-sweep first); the call performs bonds [first_bond, first_bond+max_bonds)
 * (max_bonds <= 0: to the end of the sweep) and the cache rebuilds (:770,:804) that fall inside that
 * range when rebuild_caches is set.  mode 1: construct_caches(going_left=true) only.  mode 2: the environments on
 * both sides of the current label site p (LE of the sites left of p, RE of the sites right of p) - what a sweep
 * resumed in the middle needs; with p = T-1 it is mode 1.
 * dbg: NULL or one record of dbg_stride doubles per bond performed.  seconds_out: [seconds, bonds done]. */
int orc_run(const orc_opts* o, const double* phi, const int* label, const long* counts, double* sites, long slot,
            int* chi, int* label_site, double* LE, double* RE, dgesdd_fn gesdd, int mode, int first_bond,
            int max_bonds, double* dbg, int dbg_stride, double* seconds_out) {
    orc_state st;
    memset(&st, 0, sizeof st);
    st.o = o; st.phi = phi; st.label = label; st.counts = counts; st.sites = sites; st.slot = slot; st.chi = chi;
    st.label_site = label_site; st.LE = LE; st.RE = RE; st.gesdd = gesdd;
    const int dm = o->d * o->cap;
    const long Lmax = (long)dm * dm;
    const int mmax = o->cap * o->C * o->d, nmax = dm;
    const int mnmax = mmax < nmax ? mmax : nmax;
    st.bt = malloc(sizeof(double) * Lmax * o->C);
    st.grad = malloc(sizeof(double) * Lmax * o->C);
    st.kprev = malloc(sizeof(double) * Lmax);
    st.xa = malloc(sizeof(double) * dm);
    st.xb = malloc(sizeof(double) * dm);
    st.M = malloc(sizeof(double) * (long)mmax * nmax);
    st.U = malloc(sizeof(double) * (long)mmax * mnmax);
    st.S = malloc(sizeof(double) * mnmax);
    st.Vt = malloc(sizeof(double) * (long)mnmax * nmax);
    st.lwork = 4 * mnmax * mnmax + 7 * mnmax + 3 * mnmax + (mmax > nmax ? mmax : nmax) + 4 * mnmax * mnmax + 4 * mnmax + 1024;
    st.work = malloc(sizeof(double) * st.lwork);
    st.iwork = malloc(sizeof(int) * 8 * mnmax);
    int rc = 0, done = 0;
    const int nb = o->T - 1;
    const int last = (max_bonds > 0 && first_bond + max_bonds < 2 * nb) ? first_bond + max_bonds : 2 * nb;
    const double t0 = now_s();
    if (mode == 1) {
        construct_caches(&st, 1);
    } else if (mode == 2) {
        const int p = *label_site;
        for (int j = 0; j < p && j <= o->T - 2; ++j) env_left(&st, j, st.sites + st.slot * j, st.chi[j], st.chi[j + 1]);
        for (int j = o->T - 1; j > p && j >= 1; --j) env_right(&st, j, st.sites + st.slot * j, st.chi[j], st.chi[j + 1]);
    } else {
        for (int q = first_bond; q < last && !rc; ++q) {
            if (q < nb) {
                rc = bond_step(&st, nb - 1 - q, 1, dbg ? dbg + (long)done * dbg_stride : NULL);
            } else {
                if (q == nb && o->rebuild_caches) construct_caches(&st, 0);
                rc = bond_step(&st, q - nb, 0, dbg ? dbg + (long)done * dbg_stride : NULL);
            }
            ++done;
        }
        if (last == 2 * nb && !rc && o->rebuild_caches) construct_caches(&st, 1);
    }
    if (seconds_out) { seconds_out[0] = now_s() - t0; seconds_out[1] = (double)done; }
    free(st.bt); free(st.grad); free(st.kprev); free(st.xa); free(st.xb); free(st.M); free(st.U); free(st.S);
    free(st.Vt); free(st.work); free(st.iwork);
    return rc;
}
