# Closes the TrainedMPS save / load round trip between MPSTime.jl and this package for a maintainer who has Julia
# (the build image has none).  Needs MPSTime.jl (for the JLD2 type definitions), JLD2, NPZ, SHA, ITensors.
#
#   julia --project=<MPSTime.jl checkout> roundtrip_check.jl <MPSTime.jl checkout> [model.npz written by save_trained_mps]
#
# Expected (tests/test_reference_fixture.py::REF_MPS_DIGEST, computed in the build container from the tensors
# tests/golden/extract_jld2_fixture.py read out of the same .jld2):
const EXPECTED = "0559cc1372c9561503946707a2d636d4413f8d9712b72c076c622d1619e412b6"

using MPSTime, JLD2, NPZ, SHA, ITensors        # ITensors = 0.6.22 carries MPS itself (MPSTime.jl's own pin)

# site tensor j as a plain array in the library's index order (left bond, site, right bond[, label])
function site_array(W::MPS, j::Int)
    T = length(W)
    s = siteind(W, j)
    l = j > 1 ? commonind(W[j-1], W[j]) : nothing
    r = j < T ? commonind(W[j], W[j+1]) : nothing
    lab = findindex(W[j], "f(x)")                      # src/utils.jl:342-354 looks for the same tag
    A = W[j]
    one_ind(tag) = Index(1, tag)
    if l === nothing; l = one_ind("l0"); A = A * ITensor(1.0, l); end
    if r === nothing; r = one_ind("lT"); A = A * ITensor(1.0, r); end
    return lab === nothing ? Array(A, l, s, r) : Array(A, l, s, r, lab)
end

# SHA-256 over, per site: the shape as Int64 (little endian) then the entries in C (row-major) order
function content_digest(arrs)
    ctx = SHA.SHA256_CTX()
    for A in arrs
        SHA.update!(ctx, reinterpret(UInt8, Int64.(collect(size(A)))))
        SHA.update!(ctx, reinterpret(UInt8, vec(permutedims(A, reverse(1:ndims(A))))))   # row-major bytes
    end
    return bytes2hex(SHA.digest!(ctx))
end

root = ARGS[1]
f = jldopen(joinpath(root, "test", "Data", "ecg200", "mps_saves", "test_dataset.jld2"), "r")
mps = read(f, "mps")                                    # TrainedMPS (src/Structs/options.jl:422-427)
close(f)
W = mps.mps
d_jl = content_digest([site_array(W, j) for j in 1:length(W)])
println("digest from the .jld2 through ITensors : ", d_jl, d_jl == EXPECTED ? "  OK" : "  MISMATCH")

if length(ARGS) > 1 && ARGS[2] == "--shim"
    # One command for a maintainer with Julia AND a GPU: train a small model through the shim and print the content digest of
    # the result.  `python -m mpstime_jl_amd.shim_check` (mpstime.jl_amd/shim_check.py) prints the digest of the same fit through the
    # Python mirror from the same inputs (it writes / reads shim_inputs.npz next to this file): equal digests = the ccall
    # marshalling of MPSTimeHIP.jl (layouts, 0/1-based indices, option struct) agrees with the tested binding bit for bit.
    include(joinpath(@__DIR__, "MPSTimeHIP.jl"))
    z = npzread(joinpath(@__DIR__, "shim_inputs.npz"))
    Xtr, ytr = z["X_train"], Int.(z["y_train"])
    opts = MPSOptions(; d=Int(z["d"]), chi_max=Int(z["chi_max"]), nsweeps=Int(z["nsweeps"]), eta=Float64(z["eta"]), chi_init=Int(z["chi_init"]),
                      verbosity=-1, encoding=:Legendre_No_Norm)
    # test_run = true stops after the encoding and hands back the initial MPS and the encoded states (RealRealHighDimension.jl:541, :594-597)
    W0, _, train_states, test_states, _ = fitMPS(Xtr, ytr, opts; test_run=true)
    T = size(Xtr, 2)
    # Julia's RNG stream cannot be reproduced outside Julia: the starting tensors come from the file, (left bond, site, right bond[, label])
    pos, label_idx = MPSTime.find_label(W0)
    sites = MPSTime.get_siteinds(W0)
    links = [Index(size(z["mps0_$(j-1)"], 3), "Link,l=$j") for j in 1:T-1]
    for j in 1:T
        A = z["mps0_$(j-1)"]
        is = Index[]
        j > 1 ? push!(is, links[j-1]) : (A = dropdims(A; dims=1))
        push!(is, sites[j])
        j < T ? push!(is, links[j]) : (A = dropdims(A; dims=(j > 1 ? 3 : 2)))
        j == pos && push!(is, label_idx)
        W0[j] = itensor(A, is...)
    end
    trained, info, _ = MPSTimeHIP.fitMPS_hip(W0, train_states, test_states, opts)
    println("digest of the shim-trained MPS: ", content_digest([site_array(trained.mps, j) for j in 1:T]))
    println("expected (shim_check.py):       ", read(joinpath(@__DIR__, "shim_expected_digest.txt"), String))
elseif length(ARGS) > 1
    z = npzread(ARGS[2])                                # NPZ.jl returns C-order .npy data as Julia arrays of the same shape
    T = Int(z["n_sites"])
    d_npz = content_digest([z["mps_$(j-1)"] for j in 1:T])
    println("digest from the .npz of save_trained_mps: ", d_npz, d_npz == EXPECTED ? "  OK" : "  MISMATCH")
end
