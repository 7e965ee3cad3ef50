"""Temporary instrumentation of k_imp_left (wall_clock64 per phase, written to err_out[instance 0][0..7]); apply, build,
run lab/imp_phase.py, then restore the source."""
import sys
p = 'mpstime.jl_amd/csrc/mpst_impute.hip'
s = open(p).read()
marks = [("    int seen = 0;\n    for (int step = 0; step < T; ++step) {\n        const int j = g.rev ? T - 1 - step : step;",
          "    long long tacc[8] = {0,0,0,0,0,0,0,0}; long long tlast = wall_clock64();\n#define PROF(k) do { const long long tn_ = wall_clock64(); tacc[k] += tn_ - tlast; tlast = tn_; } while (0)\n", "before"),
         ("        if (!miss) {\n            // L <- sum_s conj(phi_s) (L W)[s]", "        PROF(6);\n", "before"),
         ("            // U = LW R;  rho = U LW^H", "            PROF(0);\n", "before"),
         ("            // p_k = |rho phi_k|^2: grid values interleaved over the threads", "            PROF(1);\n", "before"),
         ("            // ---- prefix sums ----", "            PROF(2);\n", "before"),
         ("            const double* ui = g.u ? g.u + ((int64_t)i * T + j) * g.ntrial : nullptr;", "            PROF(3);\n", "before"),
         ("            if (state_from_grid) {", "            PROF(4);\n", "before"),
         ("            if (seen == nm) break;             // nothing beyond the last missing site is needed", "            PROF(5);\n", "before")]
for m, ins, _ in marks:
    i = s.index(m)
    s = s[:i] + ins + s[i:]
c = s.index("static size_t right_lds_bytes")
k = s[:c].rstrip().rfind("}")
s = s[:k] + "    if (blockIdx.x == 0 && tid == 0) for (int q = 0; q < 8; ++q) g.err_out[i * T + q] = (double)tacc[q];\n" + s[k:]
open(p, 'w').write(s)
