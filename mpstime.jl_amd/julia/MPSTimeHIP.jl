# MPSTimeHIP.jl - Julia shim that plugs libmpstime_hip.so in as the sweep engine of MPSTime.jl.
#
# The seam is the method the reference selects at src/Training/RealRealHighDimension.jl:556-560,
#     fitMPS(W::MPS, training_states_meta, testing_states_meta, opts)        (:587-890)
# next to the existing `use_legacy_ITensor` switch.  Everything above it (options, preprocessing,
# encodings, generate_startingMPS) and below it (TrainedMPS, classify) stays Julia; `impute_batch` at the end is the
# device-side body of the imputation loop (get_predictions, src/Imputation/imputation.jl:264-410).
#
# Julia is not installed in the build container or on the GPU box (SURVEY.md fact 3), so this file
# has been written to the C ABI in include/mpstime_hip.h but never executed; the Python mirror in
# mpstime.jl_amd/ drives the identical ABI and is what the tests run.  Every ITensors name used below exists in
# ITensors 0.6.22 (the reference's pin): MPS, siteinds via MPSTime.get_siteinds, linkinds, dim, array, itensor, Index.
# First run with Julia: `julia --project=<MPSTime.jl> mpstime.jl_amd/julia/roundtrip_check.jl <MPSTime.jl> --shim`
# trains one small model through this shim and prints the content digest the Python mirror prints for the same inputs.
module MPSTimeHIP

using ITensors          # the reference's only tensor dependency (src/MPSTime.jl:9, Project.toml:50 pins ITensors = 0.6.22, which still
                        # carries MPS / siteinds / linkinds itself; ITensorMPS is NOT a dependency of MPSTime.jl)
import MPSTime: EncodedTimeSeriesSet, PState, AbstractMPSOptions, MPSOptions, Options, TrainedMPS,
                safe_options, find_label, get_siteinds, KLDLoss, MSELoss, BBOpt

const LIB = get(ENV, "MPSTIME_HIP_LIB", "libmpstime_hip.so")
const MPST_ABI_VERSION = 2      # include/mpstime_hip.h this file was written against

function __init__()
    v = ccall((:mpst_version, LIB), Cint, ())
    v == MPST_ABI_VERSION || error("libmpstime_hip.so implements ABI version $v, MPSTimeHIP.jl was written for $MPST_ABI_VERSION")
end

# mpst_set_dataset's dtype: the element type everything crosses the boundary in (opts.dtype, RealRealHighDimension.jl:442)
dtype_code(::Type{Float64}) = Int32(0)
dtype_code(::Type{Float32}) = Int32(1)
dtype_code(::Type{ComplexF64}) = Int32(2)
dtype_code(::Type{ComplexF32}) = Int32(3)

struct MpstOptions            # mpst_options, field for field
    chi_max::Int32; update_iters::Int32; loss::Int32; optimiser::Int32
    rescale_before::Int32; rescale_after::Int32; train_classes_separately::Int32
    svd_alg::Int32; rebuild_caches::Int32; track_cost::Int32
    eta::Float64; cutoff::Float64
end
struct MpstSweepStats
    seconds::Float64; svd_status::Int32; max_chi::Int32; eig_sweeps_total::Int32; eig_fallbacks::Int32
end

const MPST_ERR_UNSUPPORTED = -2
const MPST_ERR_SVD = -4

"""Which librccl the library bound (it is loaded at run time, never linked): `(path_and_how, version, built_against)`."""
function comm_library()
    buf = Vector{UInt8}(undef, 1024)
    ver = Ref{Int32}(0); built = Ref{Int32}(0)
    ccall((:mpst_comm_library, LIB), Cint, (Ptr{UInt8}, Int32, Ref{Int32}, Ref{Int32}), buf, Int32(length(buf)), ver, built)
    return (unsafe_string(pointer(buf)), Int(ver[]), Int(built[]))
end

function check(ctx, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:mpst_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    rc == MPST_ERR_SVD && throw(ArgumentError(msg))          # the class tune() retries on (tuning.jl:73-86)
    rc == MPST_ERR_UNSUPPORTED && error(msg)                 # loss_functions.jl:166-170
    error("mpstime_hip [$rc]: $msg")
end

"EncodedTimeSeriesSet -> (phi[d,T,N] in the element type E, label_idx 0-based Int32)"
function pack_states(ets::EncodedTimeSeriesSet, d, T, ::Type{E}) where {E}
    N = length(ets.timeseries)
    phi = Array{E}(undef, d, T, N)
    lab = Vector{Int32}(undef, N)
    for (i, ps) in enumerate(ets.timeseries)
        for t in 1:T
            phi[:, t, i] .= ps.pstate[t]
        end
        lab[i] = Int32(ps.label_index) - 1
    end
    return phi, lab
end

"Site tensors in the boundary layout: column-major (s, l_left, l_right[, label])."
function pack_mps(W::MPS, ::Type{E}) where {E}
    T = length(W)
    pos, label_idx = find_label(W)
    sites = get_siteinds(W)
    links = linkinds(W)
    bufs = Vector{Array{E}}(undef, T)
    chi = ones(Int32, T + 1)
    for j in 1:T
        inds_j = Index[sites[j]]
        j > 1 && push!(inds_j, links[j-1]);  j > 1 && (chi[j] = dim(links[j-1]))
        j < T && push!(inds_j, links[j])
        j == pos && push!(inds_j, label_idx)
        A = E.(array(W[j], inds_j...))
        dl, dr = j > 1 ? dim(links[j-1]) : 1, j < T ? dim(links[j]) : 1
        bufs[j] = reshape(A, dim(sites[j]), dl, dr, (j == pos ? dim(label_idx) : 1))
    end
    return bufs, chi, pos - 1, label_idx, sites
end

function fitMPS_hip(W::MPS, train::EncodedTimeSeriesSet, test::EncodedTimeSeriesSet, opts::AbstractMPSOptions; device::Int=0)
    opts = safe_options(opts)
    T = length(W); d = opts.d
    E = opts.dtype                      # Float64 | Float32 | ComplexF64 | ComplexF32
    verbosity = opts.verbosity
    pos, label_idx = find_label(W)
    C = dim(label_idx)
    nsweeps = opts.nsweeps
    # a loss / an optimiser, or one per sweep (RealRealHighDimension.jl:693-713; same messages)
    if opts.loss_grad isa AbstractArray
        @assert length(opts.loss_grad) == nsweeps "loss_grad(...)::(loss,grad) must be a loss function or an array of loss functions with length nsweeps"
        loss_grads = opts.loss_grad
    elseif opts.loss_grad isa Function
        loss_grads = [opts.loss_grad for _ in 1:nsweeps]
    else
        error("loss_grad(...)::(loss,grad) must be a loss function or an array of loss functions with length nsweeps")
    end
    if opts.train_classes_separately && !(eltype(loss_grads) <: KLDLoss)                                       # :702-704
        @warn "Classes will be trained separately, but the cost function _may_ depend on measurements of multiple classes. Switch to a KLD style cost function or ensure your custom cost function depends only on one class at a time."
    end
    if opts.bbopt isa AbstractArray
        @assert length(opts.bbopt) == nsweeps "bbopt must be an optimiser or an array of optimisers to use with length nsweeps"
        bbopts = opts.bbopt
    elseif opts.bbopt isa BBOpt
        bbopts = [opts.bbopt for _ in 1:nsweeps]
    else
        error("bbopt must be an optimiser or an array of optimisers to use with length nsweeps")
    end
    loss_code(lg) = lg isa KLDLoss ? 0 : lg isa MSELoss ? 1 : -1            # -1: the library answers MPST_ERR_UNSUPPORTED
    optim_code(bb) = bb.name == "CustomGD" ? (uppercase(bb.fl) == "TSGO" ? 0 : 1) : -1
    sweep_options(its) = MpstOptions(opts.chi_max, opts.update_iters, loss_code(loss_grads[its]), optim_code(bbopts[its]), opts.rescale[1],
                                     opts.rescale[2], opts.train_classes_separately, opts.svd_alg == "recursive" ? 1 : 0, 0,
                                     opts.track_cost ? 1 : 0, opts.eta, opts.cutoff)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:mpst_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), ctx, device))
    c = ctx[]
    try
        check(c, ccall((:mpst_set_options, LIB), Cint, (Ptr{Cvoid}, Ref{MpstOptions}), c, sweep_options(1)))
        for (which, ets) in ((0, train), (1, test))
            isempty(ets.timeseries) && continue
            phi, lab = pack_states(ets, d, T, E)
            check(c, ccall((:mpst_set_dataset, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Int32}, Int64, Int32, Int32, Int32, Int32, Ptr{Int64}),
                  c, which, phi, lab, length(lab), T, d, C, dtype_code(E), C_NULL))
        end
        bufs, chi, ls, _, sites = pack_mps(W, E)
        check(c, ccall((:mpst_set_mps, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Int32}, Int32, Int32),
              c, [pointer(b) for b in bufs], chi, T, ls))
        verbosity > -1 && println("Using $(opts.update_iters) iterations per update.")                      # :629
        check(c, ccall((:mpst_build_caches, LIB), Cint, (Ptr{Cvoid},), c))

        has_test = !isempty(test.timeseries)
        info = Dict{String,Vector}("train_loss" => Float64[], "train_acc" => Float64[], "test_loss" => Float64[],
                                   "time_taken" => Float64[], "train_KL_div" => Float64[])
        has_test && merge!(info, Dict("test_acc" => Float64[], "test_KL_div" => Float64[], "test_conf" => Matrix{Float64}[]))
        function log!(t)
            opts.log_level > 0 || return NaN
            mse = Ref(0.0); kld = Ref(0.0); acc = Ref(0.0); conf = zeros(Int64, C, C)
            check(c, ccall((:mpst_eval, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ptr{Int64}), c, 0, mse, kld, acc, conf))
            push!(info["train_loss"], mse[]); push!(info["train_acc"], acc[]); push!(info["time_taken"], t); push!(info["train_KL_div"], kld[])
            verbosity > -1 && println("Training KL Div. $(kld[]) | Training acc. $(acc[]).")                  # :679-684, :835-840
            if has_test
                check(c, ccall((:mpst_eval, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ptr{Int64}), c, 1, mse, kld, acc, conf))
                push!(info["test_loss"], mse[]); push!(info["test_acc"], acc[]); push!(info["test_KL_div"], kld[])
                push!(info["test_conf"], permutedims(Float64.(conf)))     # C row-major [truth][pred] -> Julia [truth, pred]
                if verbosity > -1
                    println("Test KL Div. $(kld[]) | Testing acc. $(acc[]).")
                    println("")
                    println("Test conf: $(info["test_conf"][end]).")                                          # :685
                end
            end
            return acc[]
        end
        log!(0.0)
        nb = T - 1
        trace = zeros(Float64, opts.update_iters + 1, 2nb)
        for its in 1:nsweeps
            verbosity > -1 && println("Using optimiser $(bbopts[its].name) with the \"$(bbopts[its].fl)\" algorithm")   # :728
            verbosity > -1 && println("Starting backward sweeep: [$its/$nsweeps]")                                      # :729
            # this sweep's loss / optimiser (:727-728 index loss_grads[itS], bbopts[itS])
            its > 1 && check(c, ccall((:mpst_set_options, LIB), Cint, (Ptr{Cvoid}, Ref{MpstOptions}), c, sweep_options(its)))
            st = Ref(MpstSweepStats(0, 0, 0, 0, 0))
            check(c, ccall((:mpst_sweep, LIB), Cint, (Ptr{Cvoid}, Ref{MpstSweepStats}), c, st))
            # both half-sweeps run inside the one call: the reference's mid-sweep lines (:766, :772) go where they fall in its output -
            # between the two halves' per-bond lines when those are printed, straight after the call otherwise
            midlines() = if verbosity > -1
                println("Backward sweep finished.")                                                                    # :766
                println("Starting forward sweep: [$its/$nsweeps]")                                                     # :772
            end
            if opts.track_cost && verbosity >= 1
                # what custGD / TSGO (loss_functions.jl:50-52, :80-82) and apply_update (:181-184) print, bond by bond
                check(c, ccall((:mpst_get_loss_trace, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), c, trace))
                for q in 1:2nb
                    q == nb + 1 && midlines()
                    lid = q <= nb ? nb - q + 1 : q - nb
                    for it in 1:opts.update_iters
                        println("Loss before step $it: $(trace[it, q])")
                    end
                    println("Loss at site $lid*$(lid+1): $(trace[end, q])")
                end
            else
                midlines()
            end
            verbosity > -1 && println("Finished sweep $its. Time for sweep: $(round(st[].seconds, digits=2))s")        # :811
            acc = log!(st[].seconds)
            opts.exit_early && acc == 1.0 && break
        end
        check(c, ccall((:mpst_normalize, LIB), Cint, (Ptr{Cvoid},), c))
        verbosity > -1 && println("\nMPS normalised!\n")                                                             # :853
        log!(NaN)

        # read the MPS back into ITensors (label back on site T after the forward half-sweep)
        chi_out = zeros(Int32, T + 1); ls_out = Ref{Int32}(0)
        check(c, ccall((:mpst_get_chi, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ref{Int32}), c, chi_out, ls_out))
        outs = [Array{E}(undef, d, chi_out[j], chi_out[j+1], (j - 1 == ls_out[] ? C : 1)) for j in 1:T]
        check(c, ccall((:mpst_get_mps, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), c, [pointer(b) for b in outs]))
        links = [Index(Int(chi_out[j+1]), "Link,l=$j") for j in 1:T-1]
        Wn = MPS(T)
        for j in 1:T
            is = Index[sites[j]]
            j > 1 && push!(is, links[j-1]); j < T && push!(is, links[j]); (j - 1 == ls_out[]) && push!(is, label_idx)
            Wn[j] = itensor(reshape(outs[j], dim.(is)...), is...)
        end
        return TrainedMPS(Wn, MPSOptions(opts), train), info, test
    finally
        ccall((:mpst_destroy, LIB), Cvoid, (Ptr{Cvoid},), c)
    end
end

# ---- imputation (src/Imputation/imputation.jl:264-410: the body of get_predictions' instance loop) ----------------------
struct MpstImputeOpts            # mpst_impute_opts
    method::Int32                # 0 median, 1 mode, 2 ITS (rejection_threshold = :none), 3 mean, 4 ITS with rejection
    order::Int32                 # 0 :forwards, 1 :backwards
    get_err::Int32               # get_wmad / get_std
    max_trials::Int32
    mean_basis::Int32            # 0 :Legendre_Norm, 1 :Legendre_No_Norm, 2 :Fourier
    reserved::Int32
    rejection_threshold::Float64
end
struct MpstImputeModel           # mpst_impute_model
    N::Int64; T::Int32; d::Int32; C::Int32; label_site::Int32
    dtype::Int32                 # 0 Float64, 1 ComplexF64 (site tensors, phi and the grid states alike)
    compute::Int32               # 0 fp64, 1 fp32 chain contractions (densities stay fp64)
    site::Ptr{Ptr{Cvoid}}; chi::Ptr{Int32}; phi::Ptr{Cvoid}; label_idx::Ptr{Int32}
end

"""
    impute_batch(sites, chi, label_site, phi, label_idx, missing, xvals, xvals_enc; method, order, ...)

Every instance of `phi` ((d, T, N), encoded known values; the columns of `missing` ((T, N), UInt8) mark what is to be imputed)
with the class MPS of its label, on the GPU.  `sites[j]` is `Array(mps[j], s_j, l_{j-1}, l_j[, label])`, `xvals` /
`xvals_enc` ((d, ngrid)) are `imp.x_guess_range`.  Returns `(x, err)`, both (T, N), in the encoding's domain;
`invert_test_transform` (src/utils.jl:299) stays with the caller.
"""
function impute_batch(sites::Vector{<:Array}, chi::Vector{Int32}, label_site::Integer, phi::Array, label_idx::Vector{Int32},
                      missing::Matrix{UInt8}, xvals::Vector{Float64}, xvals_enc::Matrix;
                      method::Integer=0, order::Integer=0, get_err::Bool=true, max_trials::Integer=10, rejection_threshold::Float64=0.0,
                      u::Union{Nothing,Array{Float64}}=nothing, compute::Integer=0, device::Integer=0)
    d, T, N = size(phi)
    cx = eltype(phi) <: Complex
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:mpst_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), ctx, device))
    c = ctx[]
    try
        ptrs = [Ptr{Cvoid}(pointer(a)) for a in sites]
        x = zeros(Float64, T, N); err = zeros(Float64, T, N); secs = Ref(0.0)
        GC.@preserve sites ptrs chi phi label_idx begin
            model = Ref(MpstImputeModel(N, T, d, maximum(label_idx) + 1, label_site - 1, cx ? 1 : 0, compute,
                                        pointer(ptrs), pointer(chi), Ptr{Cvoid}(pointer(phi)), pointer(label_idx)))
            o = Ref(MpstImputeOpts(method, order, get_err ? 1 : 0, max_trials, cx ? 2 : 1, 0, rejection_threshold))
            check(c, ccall((:mpst_impute_model_run, LIB), Cint,
                           (Ptr{Cvoid}, Ref{MpstImputeModel}, Ptr{UInt8}, Ptr{Float64}, Ptr{Cvoid}, Int32, Ref{MpstImputeOpts}, Ptr{Float64},
                            Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                           c, model, missing, xvals, xvals_enc, length(xvals), o, u === nothing ? C_NULL : pointer(u), x, err, secs))
        end
        return x, err
    finally
        ccall((:mpst_destroy, LIB), Cvoid, (Ptr{Cvoid},), c)
    end
end

end # module
