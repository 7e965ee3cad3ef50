# make_reference_goldens.jl - the ONE command that pins this repository's CPU oracle (and through it the HIP engine) against
# the Julia reference itself.  The build image has no Julia, so this script has never been executed: it is written against the
# reference's source (file:line below) for a maintainer who has MPSTime.jl checked out.  Needs MPSTime, ITensors (= 0.6.22, the
# reference's pin), NPZ.
#
#   julia --project=<MPSTime.jl checkout> tests/golden/make_reference_goldens.jl <repo>/tests/golden
#
# For every fixture tests/golden/<name>.npz (encoded inputs, initial MPS, options - written by tests/golden/make_golden.py) it
# runs the REFERENCE's sweep body, bond by bond, with the reference's own functions
#   flatten_bt / apply_update / unflatten_bt / decomposeBT / update_caches! / construct_caches
#   (src/Training/RealRealHighDimension.jl:726-808)
# and writes tests/golden/juliaref_<name>.npz with the fields tests/test_oracle.py compares:
#   bond_loss, bond_grad_norm, bond_chi, bond_S (rows = bonds in sweep order, zero padded), train_KL_div (before the first and after
#   every sweep).  tests/test_oracle.py::test_oracle_against_reference_vectors picks the files up when they exist.
# Two more fixtures (tests/golden/make_golden_complex_impute.py) are run at the end of the script:
#   complex_kld_c2.npz    through the reference's LEGACY ITensor engine (use_legacy_ITensor = true: the only engine of the reference that
#                         takes a complex encoding, RealRealHighDimension.jl:461-466) -> juliaref_complex_kld_c2.npz
#                         (tests/test_oracle_complex.py::test_complex_oracle_against_reference_vectors)
#   impute_median_c1.npz  one instance through impute_median(...; get_wmad = true), both orders -> juliaref_impute_median_c1.npz
#                         (tests/test_impute_oracle.py::test_imputation_oracle_against_reference_vectors)
using MPSTime, ITensors, NPZ, LinearAlgebra
import MPSTime: PState, EncodedTimeSeriesSet, TrainSeparate, flatten_bt, unflatten_bt, apply_update, decomposeBT, update_caches!,
                construct_caches, MSE_loss_acc, safe_options, find_label

dir = ARGS[1]
# NPZ.jl reads numeric arrays only: the two string options of a fixture are repeated here (tests/golden/make_golden.py: CASES)
const LOSS_BBOPT = Dict("kld_tsgo_c2" => (:KLD, :TSGO), "kld_sep_c3_ragged" => (:KLD, :TSGO), "mse_gd_c2_iters3" => (:MSE, :GD),
                        "kld_c1_unsupervised" => (:KLD, :TSGO), "two_site_mps" => (:KLD, :TSGO), "config1_trendy_sine" => (:KLD, :TSGO))
skip(p) = any(startswith(basename(p), pre) for pre in ("ref_", "juliaref_", "complex_", "impute_"))
for path in sort(filter(p -> endswith(p, ".npz") && !skip(p), readdir(dir; join=true)))
    name = splitext(basename(path))[1]
    haskey(LOSS_BBOPT, name) || (println("skipping ", name, ": not in LOSS_BBOPT"); continue)
    T = size(npzread(path, ["phi"])["phi"], 2)
    g = npzread(path, vcat(["phi", "label_index", "class_distribution", "opts", "eta"], ["W0_$(j-1)" for j in 1:T]))
    phi = g["phi"]                                      # (N, T, d)
    N, T, d = size(phi)
    lab = Int.(g["label_index"])                        # 0-based class slot, non-decreasing
    chi_max, iters, nsweeps, sep = Int.(g["opts"])
    eta = Float64(first(g["eta"])); loss, bb = LOSS_BBOPT[name]
    C = length(g["class_distribution"])
    opts = safe_options(MPSOptions(; d=d, chi_max=chi_max, eta=eta, nsweeps=nsweeps, update_iters=iters, loss_grad=loss, bbopt=bb,
                                   train_classes_separately=(sep != 0), cutoff=1e-10, rescale=(false, true), verbosity=-1, log_level=0,
                                   encoding=:Legendre_No_Norm, dtype=Float64))
    # product states (src/Structs/structs.jl:12-17): pstate[t] = the d values of site t
    states = [PState([phi[i, t, :] for t in 1:T], lab[i], UInt(lab[i] + 1)) for i in 1:N]
    ets = EncodedTimeSeriesSet(states, zeros(N, T), Int.(g["class_distribution"]))
    # the initial MPS from the fixture: W0_j has axes (left bond, site, right bond[, label]) (oracle/ref_numpy.py random_mps)
    sites = siteinds(d, T)                              # RealRealHighDimension.jl:431
    label_idx = Index(C, "f(x)")                        # :19
    W0 = [g["W0_$(j-1)"] for j in 1:T]
    links = [Index(size(W0[j], 3), "Link,l=$j") for j in 1:T-1]
    W = MPS(T)
    for j in 1:T
        A = W0[j]
        is = Index[]
        j > 1 && push!(is, links[j-1]); push!(is, sites[j]); j < T && push!(is, links[j]); ndims(A) == 4 && push!(is, label_idx)
        keep = [j > 1 ? Colon() : 1, Colon(), j < T ? Colon() : 1]
        ndims(A) == 4 && push!(keep, Colon())
        W[j] = itensor(A[keep...], is...)
    end
    tsep = TrainSeparate{opts.train_classes_separately}()                                  # :606
    dtype = opts.dtype
    LE, RE = construct_caches(W, states; going_left=true, dtype=dtype)                      # :631
    bond_loss = Float64[]; bond_grad = Float64[]; bond_chi = Int64[]; bond_S = Vector{Float64}[]
    klds = Float64[MSE_loss_acc(W, states)[2]]                                             # :660
    function bond!(j, going_left)
        bt, bt_inds = flatten_bt(W[j], W[j+1], label_idx, dtype; going_left=going_left)    # :733 / :777
        l, gr = opts.loss_grad(tsep, bt, LE, RE, ets, j, j + 1)                            # what custGD / TSGO evaluate first (loss_functions.jl:44,75)
        push!(bond_loss, l); push!(bond_grad, norm(gr))
        bt_new = apply_update(tsep, bt, LE, RE, j, j + 1, ets; iters=opts.update_iters, verbosity=-1, dtype=dtype, loss_grad=opts.loss_grad,
                              bbopt=opts.bbopt, track_cost=false, eta=opts.eta, rescale=opts.rescale)          # :736 / :779
        bt_it = unflatten_bt(bt_new, bt_inds)
        lsn, rsn = decomposeBT(bt_it, j, j + 1; chi_max=opts.chi_max, cutoff=opts.cutoff, going_left=going_left, dtype=dtype, alg=opts.svd_alg)   # :756 / :798
        update_caches!(lsn, rsn, LE, RE, j, j + 1, states; going_left=going_left)           # :759 / :799
        W[j] = lsn; W[j+1] = rsn
        k = dim(commonind(lsn, rsn))
        _, S, _ = svd(lsn * rsn, uniqueinds(lsn, rsn))                                      # gauge-invariant: the kept singular values
        push!(bond_chi, k); push!(bond_S, sort(diag(Array(S, inds(S)...)); rev=true)[1:k])
    end
    for its in 1:nsweeps
        for j in (T-1):-1:1; bond!(j, true); end
        LE, RE = construct_caches(W, states; going_left=false)                             # :770
        for j in 1:(T-1); bond!(j, false); end
        LE, RE = construct_caches(W, states; going_left=true)                              # :804
        push!(klds, MSE_loss_acc(W, states)[2])
    end
    smax = maximum(length.(bond_S))
    Smat = zeros(length(bond_S), smax)
    for (i, s) in enumerate(bond_S); Smat[i, 1:length(s)] .= s; end
    out = joinpath(dir, "juliaref_" * basename(path))
    npzwrite(out, Dict("bond_loss" => bond_loss, "bond_grad_norm" => bond_grad, "bond_chi" => bond_chi, "bond_S" => Smat, "train_KL_div" => klds))
    println("wrote ", out, ": ", length(bond_loss), " bonds")
end


# ---------------------------------------------------------------------------------------------------------------------------------
# The complex fixture through the legacy ITensor engine: the sweep body of fitMPS_IT (src/legacy_itensor/RealRealLegacyITensor.jl:280-365)
# with the reference's own apply_update_IT / decomposeBT_IT / update_caches_IT! / construct_caches_IT, bond by bond.
import MPSTime: PStateIT, EncodedTimeSeriesSetIT, apply_update_IT, decomposeBT_IT, update_caches_IT!, construct_caches_IT
let path = joinpath(dir, "complex_kld_c2.npz")
    T = size(npzread(path, ["phi"])["phi"], 2)
    g = npzread(path, vcat(["phi", "label_index", "class_distribution", "opts", "eta"], ["W0_$(j-1)" for j in 1:T]))
    phi = g["phi"]                                      # (N, T, d) ComplexF64
    N, T, d = size(phi)
    lab = Int.(g["label_index"])
    chi_max, iters, nsweeps, sep = Int.(g["opts"])
    C = length(g["class_distribution"])
    opts = safe_options(MPSOptions(; d=d, chi_max=chi_max, eta=Float64(first(g["eta"])), nsweeps=nsweeps, update_iters=iters, loss_grad=:KLD,
                                   bbopt=:TSGO, train_classes_separately=(sep != 0), cutoff=1e-10, rescale=(false, true), verbosity=-1,
                                   log_level=0, encoding=:Fourier, dtype=ComplexF64, use_legacy_ITensor=true))
    sites = siteinds(d, T)
    label_idx = Index(C, "f(x)")
    states = [PState([phi[i, t, :] for t in 1:T], lab[i], UInt(lab[i] + 1)) for i in 1:N]
    ets = EncodedTimeSeriesSetIT(EncodedTimeSeriesSet(states, zeros(N, T), Int.(g["class_distribution"])), sites)   # structs.jl:57-61
    states_it = ets.timeseries
    W0 = [g["W0_$(j-1)"] for j in 1:T]
    links = [Index(size(W0[j], 3), "Link,l=$j") for j in 1:T-1]
    W = MPS(T)
    for j in 1:T
        A = W0[j]
        is = Index[]
        j > 1 && push!(is, links[j-1]); push!(is, sites[j]); j < T && push!(is, links[j]); ndims(A) == 4 && push!(is, label_idx)
        keep = [j > 1 ? Colon() : 1, Colon(), j < T ? Colon() : 1]
        ndims(A) == 4 && push!(keep, Colon())
        W[j] = itensor(A[keep...], is...)
    end
    tsep = TrainSeparate{opts.train_classes_separately}()
    dtype = opts.dtype
    LE, RE = construct_caches_IT(W, states_it; going_left=true, dtype=dtype)                                       # :191
    bond_loss = Float64[]; bond_grad = Float64[]; bond_chi = Int64[]; bond_S = Vector{Float64}[]
    klds = Float64[MSE_loss_acc(W, states_it)[2]]
    function bond_it!(j, going_left)
        BT = going_left ? W[j+1] * W[j] : W[j] * W[j+1]                                                            # :293 / :332
        l, gr = opts.loss_grad(tsep, BT, LE, RE, ets, j, j + 1)                    # what custGD / TSGO evaluate first (legacy loss_functions.jl:108-170)
        push!(bond_loss, l); push!(bond_grad, norm(gr))
        BT_new = apply_update_IT(tsep, BT, LE, RE, j, j + 1, ets; iters=opts.update_iters, verbosity=-1, dtype=dtype, loss_grad=opts.loss_grad,
                                 bbopt=opts.bbopt, track_cost=false, eta=opts.eta, rescale=opts.rescale)           # :295-311 / :334-350
        lsn, rsn = decomposeBT_IT(BT_new, j, j + 1; chi_max=opts.chi_max, cutoff=opts.cutoff, going_left=going_left, dtype=dtype, alg=opts.svd_alg)
        update_caches_IT!(lsn, rsn, LE, RE, j, j + 1, states_it; going_left=going_left)                            # :317 / :353
        W[j] = lsn; W[j+1] = rsn
        k = dim(commonind(lsn, rsn))
        _, S, _ = svd(lsn * rsn, uniqueinds(lsn, rsn))
        push!(bond_chi, k); push!(bond_S, sort(real.(diag(Array(S, inds(S)...))); rev=true)[1:k])
    end
    for its in 1:nsweeps
        for j in (T-1):-1:1; bond_it!(j, true); end
        LE, RE = construct_caches_IT(W, states_it; going_left=false)                                               # :328
        for j in 1:(T-1); bond_it!(j, false); end
        LE, RE = construct_caches_IT(W, states_it; going_left=true)                                                # :358
        push!(klds, MSE_loss_acc(W, states_it)[2])
    end
    smax = maximum(length.(bond_S))
    Smat = zeros(length(bond_S), smax)
    for (i, s) in enumerate(bond_S); Smat[i, 1:length(s)] .= s; end
    out = joinpath(dir, "juliaref_complex_kld_c2.npz")
    npzwrite(out, Dict("bond_loss" => bond_loss, "bond_grad_norm" => bond_grad, "bond_chi" => bond_chi, "bond_S" => Smat, "train_KL_div" => klds))
    println("wrote ", out, ": ", length(bond_loss), " bonds (legacy ITensor engine, ComplexF64)")
end

# ---------------------------------------------------------------------------------------------------------------------------------
# One imputation instance through the reference's impute_median (src/Imputation/MPS_methods.jl:201-230): precondition on the known
# values, orthogonalise onto the first missing site, walk the missing sites (median of the conditional density on the grid, weighted
# median absolute deviation), both imputation orders.
import MPSTime: impute_median, EncodedDataRange
let path = joinpath(dir, "impute_median_c1.npz")
    T = length(npzread(path, ["x"])["x"])
    g = npzread(path, vcat(["x", "enc", "missing", "xs", "grid_phi"], ["mps_$(j-1)" for j in 1:T]))
    x = Float64.(g["x"]); enc = g["enc"]; xs = Float64.(g["xs"]); gphi = g["grid_phi"]
    d = size(enc, 2)
    missing_sites = Int.(g["missing"]) .+ 1                                       # 1-based
    opts = safe_options(MPSOptions(; d=d, encoding=:Legendre_No_Norm, verbosity=-1, log_level=0))
    sites = siteinds(d, T)
    A = [g["mps_$(j-1)"] for j in 1:T]                                            # (Dl, d, Dr), Dl = 1 on the first, Dr = 1 on the last site
    links = [Index(size(A[j], 3), "Link,l=$j") for j in 1:T-1]
    class_mps = MPS(T)
    for j in 1:T
        is = Index[]
        j > 1 && push!(is, links[j-1]); push!(is, sites[j]); j < T && push!(is, links[j])
        class_mps[j] = itensor(A[j][j > 1 ? Colon() : 1, :, j < T ? Colon() : 1], is...)
    end
    ts_enc = MPS([itensor(enc[t, :], sites[t]) for t in 1:T])                     # the encoded series as a product state
    single = [gphi[k, :] for k in 1:length(xs)]                                   # imputation.jl:101-107: one table, a view per site
    rng_enc = EncodedDataRange(xs[2] - xs[1], (xs[1], xs[end]), xs, sites[1], [view(single, :) for _ in 1:T])
    res = Dict{String,Any}()
    for order in (:forwards, :backwards)
        xi, wmad = impute_median(class_mps, opts, rng_enc, [], x, ts_enc, missing_sites; impute_order=order, get_wmad=true)
        res["x_$(order)"] = xi[missing_sites]
        res["wmad_$(order)"] = wmad[missing_sites]
    end
    out = joinpath(dir, "juliaref_impute_median_c1.npz")
    npzwrite(out, res)
    println("wrote ", out)
end
