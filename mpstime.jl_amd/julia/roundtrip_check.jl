# Closes the TrainedMPS save / load round trip between MPSTime.jl and this package for a maintainer who has Julia
# (the build image has none).  Needs MPSTime.jl (for the JLD2 type definitions), JLD2, NPZ, SHA, ITensors.
#
#   julia --project=<MPSTime.jl checkout> roundtrip_check.jl <MPSTime.jl checkout> [model.npz written by save_trained_mps]
#
# Expected (tests/test_reference_fixture.py::REF_MPS_DIGEST, computed in the build container from the tensors
# tests/golden/extract_jld2_fixture.py read out of the same .jld2):
const EXPECTED = "0559cc1372c9561503946707a2d636d4413f8d9712b72c076c622d1619e412b6"

using MPSTime, JLD2, NPZ, SHA, ITensors, ITensorMPS

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

if length(ARGS) > 1
    z = npzread(ARGS[2])                                # NPZ.jl returns C-order .npy data as Julia arrays of the same shape
    T = Int(z["n_sites"])
    d_npz = content_digest([z["mps_$(j-1)"] for j in 1:T])
    println("digest from the .npz of save_trained_mps: ", d_npz, d_npz == EXPECTED ? "  OK" : "  MISMATCH")
end
