# Monodepth2HIP.jl — the Julia `ccall` shim over libmd2hip.so (include/md2hip.h).
#
# NOT EXECUTED IN THIS REPO'S CI: the build image has no Julia.  The mechanical half is GENERATED from the same tables the Python binding and the
# tests use (md2_ccall.jl <- _lib.py:_SIGS, param_table.jl <- synthetic.py:param_layout; tests/test_julia_shim.py keeps both in sync and checks that
# every export is bound here); this file is what a maintainer of pxl-th/Monodepth2.jl adds to keep the module's call surface (src/Monodepth.jl,
# model.jl, training.jl, utils.jl) while the training step runs on an MI355X.  Device memory is owned by AMDGPU.jl `ROCArray{Float32}`s (any array
# type exposing a device pointer works); Julia's column-major (W,H,C,N) bytes are passed unchanged.
module Monodepth2HIP

using AMDGPU
using LinearAlgebra: I
import ChainRulesCore
import ChainRulesCore: @non_differentiable, NoTangent, unthunk

const LIB = get(ENV, "MD2HIP_LIB", joinpath(@__DIR__, "..", "libmd2hip.so"))

struct MD2Config
    batch::Int32; channels::Int32; height::Int32; width::Int32
    automasking::Int32; device::Int32
    min_depth::Float64; max_depth::Float64; disparity_smoothness::Float64; lr::Float64
    encoder_depth::Int32; precision::Int32       # ResidualNetwork(18 | 50); 0 = f32, 1 = bf16 MFMA operands on fp32 tensors, 2 = bf16 activation tensors
    target_id::Int32; source_id1::Int32; source_id2::Int32      # TrainCache.target_id / source_ids (1-based; 0 = the defaults 2 / (1, 3))
    scale_levels::NTuple{5, Int32}                               # DepthDecoder(; scale_levels), ascending, 0-terminated (all 0 = 2:5, src/Monodepth.jl:106)
    scales::NTuple{5, Float64}                                   # TrainCache.scales: ALL five 0 = the defaults 1 / 2^(5 - level) (src/Monodepth.jl:107); otherwise as given (0 = that output's smoothness off, negative = its default)
end

include("md2_ccall.jl")        # raw wrappers: md2_create(...), md2_train_step(...), ... one per export
include("param_table.jl")      # PARAM_TABLE[(C, depth)]: the flat parameter order

mutable struct Handle
    ptr::Ptr{Cvoid}
    channels::Int
    depth::Int
    scale_levels::Vector{Int}
end
pad5(v, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), 5)
disp5(d) = ntuple(i -> i <= length(d) ? dptr(d[i]) : C_NULL, 5)      # the C ABI takes one pointer per possible scale level

function check(h, rc, what)
    rc == 0 && return
    msg = unsafe_string(md2_last_error(h === nothing ? C_NULL : h.ptr))
    error("libmd2hip $what failed ($rc): $msg")          # reference convention: plain Julia exceptions
end

dptr(x::ROCArray{Float32}) = Ptr{Cvoid}(UInt(pointer(x)))
dptr(::Nothing) = C_NULL
hptr(x::Array) = Ptr{Cvoid}(pointer(x))
const NOSTREAM = C_NULL

# ---- a18: the reference's own device seam (src/utils.jl:1-11), for the device array type of this backend ----------------------------------
eye_like(::ROCArray{T}, shape) where T = ROCArray{T, 2}(Matrix{T}(I, shape))
zeros_like(::ROCArray{T}, shape) where T = AMDGPU.zeros(T, shape)
ones_like(::ROCArray{T}, shape) where T = AMDGPU.ones(T, shape)
@non_differentiable eye_like(::Any...)
@non_differentiable zeros_like(::Any...)
@non_differentiable ones_like(::Any...)

# ---- handle: Params (src/Monodepth.jl:32-42) + the shapes TrainCache is built for ------------------------------------------------------------
function Handle(; batch, channels, width, height, automasking=false, device=0,
                min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, lr=1e-4, encoder_depth=18, precision=0, target_id=2, source_ids=[1, 3],
                scale_levels=collect(2:5), scales=Float64[])
    # md2_config.scales: an all-zero array means "not given".  A caller whose scales are ALL 0 wants every smoothness weight disparity_smoothness * scale_i (src/training.jl:67)
    # to be 0: say that through the other factor, as the Python mirror does (monodepth.py, Model._ensure)
    if !isempty(scales) && all(iszero, scales)
        scales = Float64[]; disparity_smoothness = 0.0
    end
    cfg = Ref(MD2Config(batch, channels, height, width, automasking, device, min_depth, max_depth, disparity_smoothness, lr, encoder_depth, precision,
                        target_id, source_ids[1], source_ids[2],           # TrainCache.target_id / source_ids (src/Monodepth.jl:44-55), 1-based like the reference
                        pad5(scale_levels, Int32), pad5(scales, Float64)))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve out begin
        rc = md2_create(cfg, Base.unsafe_convert(Ptr{Ptr{Cvoid}}, out))
    end
    check(nothing, rc, "md2_create")
    h = Handle(out[], channels, encoder_depth, collect(Int, scale_levels)); finalizer(x -> md2_destroy(x.ptr), h); h
end
set_stream!(h, stream::Ptr{Cvoid}) = check(h, md2_set_stream(h.ptr, stream), "set_stream")
synchronize(h) = check(h, md2_synchronize(h.ptr), "synchronize")
# TrainCache.K / invK (src/Monodepth.jl:113): 3x3 column-major, host
set_intrinsics!(h, K::Matrix{Float32}, invK::Matrix{Float32}) = GC.@preserve K invK check(h, md2_set_intrinsics(h.ptr, hptr(K), hptr(invK)), "set_intrinsics")
set_lr!(h, eta) = check(h, md2_set_lr(h.ptr, Float64(eta)), "set_lr")
testmode!(h, on::Bool=true) = check(h, md2_set_testmode(h.ptr, Int32(on)), "set_testmode")       # Flux.testmode! / trainmode!, src/Monodepth.jl:126,209,237
trainmode!(h) = testmode!(h, false)
set_deterministic!(on::Bool) = md2_set_deterministic(Int32(on))
# test switch: the op-level convolutions run the bf16-tensor kernels of precision = 2
set_op_storage!(bf16::Bool) = md2_set_op_storage(Int32(bf16))
set_winograd!(on::Union{Bool,Integer}, min_blocks::Integer=-1) = md2_set_winograd(Int32(on), Int32(min_blocks))     # algorithm choice of the 3x3 / stride-1 convolutions (cuDNN's job behind Flux `Conv`): 0 direct, 1 automatic, 2 / 3 pin a Winograd kernel
# how the convolutions form their fp32 products (cuDNN's math mode): 0 fp32 MFMA, 3 exact three-bf16-piece split, 1 (default) the faster of the two per launch, 2 reduced precision (opt-in); -1 keeps
set_conv_products!(fwd_dgrad::Integer=-1, wgrad::Integer=-1) = md2_set_conv_products(Int32(fwd_dgrad), Int32(wgrad))
function get_conv_products()
    f = Ref(Int32(0)); w = Ref(Int32(0))
    GC.@preserve f w md2_get_conv_products(Base.unsafe_convert(Ptr{Int32}, f), Base.unsafe_convert(Ptr{Int32}, w))
    return (Int(f[]), Int(w[]))
end
set_two_streams!(h, on::Bool) = check(h, md2_set_two_streams(h.ptr, Int32(on)), "set_two_streams")

# ---- Flux.params(model) <-> the library's flat vector ----------------------------------------------------------------------------------------
params_count(h) = md2_params_count(h.ptr)
load_params!(h, θ::Vector{Float32}) = GC.@preserve θ check(h, md2_params_set(h.ptr, hptr(θ), length(θ)), "params_set")
function params(h); θ = Vector{Float32}(undef, params_count(h)); GC.@preserve θ check(h, md2_params_get(h.ptr, hptr(θ), length(θ)), "params_get"); θ; end
function grads(h); g = Vector{Float32}(undef, params_count(h)); GC.@preserve g check(h, md2_grads_get(h.ptr, hptr(g), length(g)), "grads_get"); g; end
function bn_state(h); s = Vector{Float32}(undef, md2_bn_state_count(h.ptr)); GC.@preserve s check(h, md2_bn_state_get(h.ptr, hptr(s), length(s)), "bn_state_get"); s; end
load_bn_state!(h, s::Vector{Float32}) = GC.@preserve s check(h, md2_bn_state_set(h.ptr, hptr(s), length(s)), "bn_state_set")
function opt_state(h)
    n = params_count(h); m = Vector{Float32}(undef, n); v = Vector{Float32}(undef, n); b1 = Ref(0.0); b2 = Ref(0.0)
    GC.@preserve m v b1 b2 check(h, md2_opt_state_get(h.ptr, hptr(m), hptr(v), n, Base.unsafe_convert(Ptr{Float64}, b1), Base.unsafe_convert(Ptr{Float64}, b2)), "opt_state_get")
    m, v, b1[], b2[]
end
load_opt_state!(h, m::Vector{Float32}, v::Vector{Float32}, b1, b2) = GC.@preserve m v check(h, md2_opt_state_set(h.ptr, hptr(m), hptr(v), length(m), Float64(b1), Float64(b2)), "opt_state_set")

# The trainable arrays of `Model(encoder, depth_decoder, pose_decoder)` (src/model.jl:1-6) in the library's flat order (param_table.jl).
#  * depth_decoder.branches is a Vector of Vectors of BranchBlock (src/depth_decoder.jl:36-44): branches 1..5 in order, each c1.d.weight, c1.d.bias, c2.d.weight,
#    c2.d.bias; then depth_decoder.decoders[l].d.(weight, bias) for the scale levels (2..5 by default); pose_decoder.squeezer, pose_decoder.pose[1..3] (src/pose_decoder.jl:13-21).
#    This is exactly the order `Flux.params` walks these structs in.
#  * encoder: ResNet.jl is not vendored with the reference, so its field names are not relied on: `Flux.params(m.encoder)` is walked in order and every array's
#    size is checked against the table.  Flux's BatchNorm yields (β, γ) — `trainable(bn) = (β, γ)` — while the table holds (γ, β): the pair is swapped here.
#    A ResNet.jl that orders its layers differently (e.g. the shortcut before conv1) fails the size check with the entry it stopped at; adapt the walk there.
# The table for a DepthDecoder built with other scale_levels (src/depth_decoder.jl:38-46): branches 1..max level, one head per level (decoder_channels[level] => 1),
# offsets recomputed.  param_table.jl holds the default 2:5.
function param_table(channels::Int, depth::Int, scale_levels=2:5)
    base = PARAM_TABLE[(channels, depth)]
    collect(scale_levels) == collect(2:5) && return base
    dec = (256, 128, 64, 32, 16)
    rows = Tuple{String,Symbol,Tuple}[]
    for (name, kind, shape, _) in base
        startswith(name, "pose.") && continue
        startswith(name, "dec.head") && continue
        m = match(r"^dec\.branch(\d)", name)
        m !== nothing && parse(Int, m.captures[1]) > maximum(scale_levels) && continue
        push!(rows, (name, kind, shape))
    end
    for l in scale_levels
        push!(rows, ("dec.head$l.W", :W, (3, 3, dec[l], 1))); push!(rows, ("dec.head$l.b", :b, (1,)))
    end
    for (name, kind, shape, _) in base
        startswith(name, "pose.") && push!(rows, (name, kind, shape))
    end
    out = Tuple{String,Symbol,Tuple,Int}[]; off = 0
    for (name, kind, shape) in rows
        push!(out, (name, kind, shape, off)); off += prod(shape)
    end
    out
end
function ordered_params(m, channels::Int, depth::Int, params_of=x -> collect(Flux_params(x)); scale_levels=2:5)
    tab = param_table(channels, depth, scale_levels)
    out = Vector{Any}(undef, length(tab))
    enc = params_of(m.encoder); ei = 1; i = 1
    while i <= length(tab) && startswith(tab[i][1], "enc.")
        if tab[i][2] == :gamma                         # table: gamma, beta  <-  Flux: β, γ
            β, γ = enc[ei], enc[ei + 1]; ei += 2
            size(γ) == tab[i][3] && size(β) == tab[i + 1][3] || error("encoder parameter $(ei - 2): expected a BatchNorm of size $(tab[i][3]) at $(tab[i][1]), got $(size(β)), $(size(γ))")
            out[i] = γ; out[i + 1] = β; i += 2
        else
            size(enc[ei]) == tab[i][3] || error("encoder parameter $ei: expected $(tab[i][1]) of size $(tab[i][3]), got $(size(enc[ei]))")
            out[i] = enc[ei]; ei += 1; i += 1
        end
    end
    ei - 1 == length(enc) || error("the encoder has $(length(enc)) trainable arrays, the table uses $(ei - 1) (a classifier head? build ResidualNetwork(n; classes=nothing), src/Monodepth.jl:117)")
    rest = Any[]
    for branch in m.depth_decoder.branches, b in branch
        push!(rest, b.c1.d.weight, b.c1.d.bias, b.c2.d.weight, b.c2.d.bias)
    end
    for d in m.depth_decoder.decoders
        push!(rest, d.d.weight, d.d.bias)
    end
    push!(rest, m.pose_decoder.squeezer.weight, m.pose_decoder.squeezer.bias)
    for l in m.pose_decoder.pose
        push!(rest, l.weight, l.bias)
    end
    length(rest) == length(tab) - i + 1 || error("decoder parameter count mismatch")
    for (k, a) in enumerate(rest)
        size(a) == tab[i + k - 1][3] || error("expected $(tab[i + k - 1][1]) of size $(tab[i + k - 1][3]), got $(size(a))")
        out[i + k - 1] = a
    end
    out
end
Flux_params(x) = Main.Flux.params(x)          # resolved in the host module, which has Flux loaded (src/Monodepth.jl:13)

# Flux.params(model) -> the flat host vector md2_params_set takes (src/Monodepth.jl:124)
flatten_flux_params(m, channels, depth=18; scale_levels=2:5) = reduce(vcat, (vec(Float32.(Array(a))) for a in ordered_params(m, channels, depth; scale_levels)))
# the reverse, in place, for `model_host = cpu(model); @save ... model_host` (src/Monodepth.jl:191-194)
function unflatten_flux_params!(m, θ::Vector{Float32}, channels, depth=18; scale_levels=2:5)
    tab = param_table(channels, depth, scale_levels)
    for (a, (_, _, shape, off)) in zip(ordered_params(m, channels, depth; scale_levels), tab)
        copyto!(a, reshape(θ[off + 1:off + prod(shape)], shape))
    end
    m
end
# BatchNorm μ / σ² (part of what the BSON checkpoint keeps): md2_bn_state_get order = the table's BatchNorms in order, mean then variance.  `bns` = the encoder's
# BatchNorm layers in the same order (e.g. `[l for l in Flux.modules(m.encoder) if l isa BatchNorm]`).
function unflatten_bn_state!(bns, s::Vector{Float32})
    o = 0
    for bn in bns
        c = length(bn.μ); copyto!(bn.μ, s[o + 1:o + c]); copyto!(bn.σ², s[o + c + 1:o + 2c]); o += 2c
    end
    o == length(s) || error("BatchNorm state size mismatch")
end

# ---- model / loss / step ------------------------------------------------------------------------------------------------------------------------
# (m::Model)(x, source_ids, target_id), src/model.jl:8-20.  x: ROCArray (W,H,C,3,N)
function model_forward(h, x::ROCArray{Float32,5})
    W, H, C, L, N = size(x)
    disps = [ROCArray{Float32}(undef, W >> (5 - l), H >> (5 - l), 1, N) for l in h.scale_levels]
    rvec = ROCArray{Float32}(undef, 3, N, 2); tvec = ROCArray{Float32}(undef, 3, N, 2)       # Julia (3,N,2) == C [2][N][3]
    check(h, md2_model_forward(h.ptr, dptr(x), disp5(disps)..., dptr(rvec), dptr(tvec)), "model_forward")
    disps, rvec, tvec
end
# train_loss(model, x, auto_loss, cache, parameters, false)[1], src/training.jl:21-78 -> device scalar
# eval_poses(m, features, source_ids, target_id), src/model.jl:22-35, from the frames (the stage features live in the handle): encoder + pose decoder only
function eval_poses(h, x::ROCArray{Float32,5})
    N = size(x, 5); rvec = ROCArray{Float32}(undef, 3, N, 2); tvec = ROCArray{Float32}(undef, 3, N, 2)
    check(h, md2_eval_poses(h.ptr, dptr(x), dptr(rvec), dptr(tvec)), "eval_poses"); rvec, tvec
end
function train_loss(h, x::ROCArray{Float32,5}, auto_loss)
    loss = ROCArray{Float32}(undef, 1)
    check(h, md2_train_loss(h.ptr, dptr(x), dptr(auto_loss), dptr(loss)), "train_loss"); loss
end
# gradient(θ) do train_loss(...) end  — as a ChainRulesCore.rrule the pullback is `Δ -> (md2_backward; grads(h) .* Δ)`
backward!(h) = check(h, md2_backward(h.ptr), "backward")
# Flux.Optimise.update!(ADAM(1e-4), θ, ∇), src/Monodepth.jl:170
adam_step!(h, grad_scale=1f0) = check(h, md2_adam_step(h.ptr, Float32(grad_scale)), "adam_step")
# the loop body src/Monodepth.jl:156-176 in one call (with a communicator attached: + all-reduce, ADAM on the mean)
function train_step!(h, x, auto_loss=nothing; update=true)
    loss = ROCArray{Float32}(undef, 1)
    check(h, md2_train_step(h.ptr, dptr(x), dptr(auto_loss), dptr(loss), Int32(update)), "train_step"); loss
end
# automasking_loss(ssim, x, x[:,:,:,target,:]; source_ids), src/training.jl:9-11
function automasking_loss(h, x::ROCArray{Float32,5})
    W, H, _, _, N = size(x); out = ROCArray{Float32}(undef, W, H, 1, N)
    check(h, md2_automasking_loss(h.ptr, dptr(x), dptr(out)), "automasking_loss"); out
end
# eval_disparity(m, x), src/model.jl:28.  x: ROCArray (W,H,C,n) single frames, n <= batch
function eval_disparity(h, x::ROCArray{Float32,4})
    W, H, C, n = size(x)
    disps = [ROCArray{Float32}(undef, W >> (5 - l), H >> (5 - l), 1, n) for l in h.scale_levels]
    check(h, md2_eval_disparity(h.ptr, dptr(x), Int32(n), disp5(disps)...), "eval_disparity"); disps
end
# the loss half on given disparities / poses and its pullbacks (src/training.jl:28-77)
function loss_fwd_bwd(h, x, disps, rvec, tvec, auto_loss=nothing)
    loss = ROCArray{Float32}(undef, 1); dd = similar.(disps); dr = similar(rvec); dt = similar(tvec)
    check(h, md2_loss_fwd_bwd(h.ptr, dptr(x), disp5(disps)..., dptr(rvec), dptr(tvec), dptr(auto_loss), dptr(loss), disp5(dd)..., dptr(dr), dptr(dt)), "loss_fwd_bwd")
    loss, dd, dr, dt
end
# slow_depth's objective + pullbacks (src/simple_depth.jl:25-42): disp (1,W,H,N) == [N][1][H][W]
function simple_depth_loss(h, x, disp, rvec, tvec)
    loss = ROCArray{Float32}(undef, 1); dd = similar(disp); dr = similar(rvec); dt = similar(tvec)
    check(h, md2_simple_depth_loss(h.ptr, dptr(x), dptr(disp), dptr(rvec), dptr(tvec), dptr(loss), dptr(dd), dptr(dr), dptr(dt)), "simple_depth_loss")
    loss, dd, dr, dt
end

# ---- data parallelism: the library's own RCCL communicator (one Julia process per GPU; the reference is single-GPU) ---------------------------------
# rank 0: id = comm_unique_id(); ship the 128 bytes to the other ranks (MPI.bcast, a file, a TCP store ...); every rank: comm_init!(h, rank, world, id);
# broadcast_state!(h) makes the replicas identical; after that train_step! all-reduces inside md2_backward and applies ADAM to the mean.
function comm_unique_id(); id = Vector{UInt8}(undef, 128); GC.@preserve id (rc = md2_comm_unique_id(hptr(id)); rc == 0 || error("md2_comm_unique_id failed ($rc)")); id; end
comm_init!(h, rank, world, id::Vector{UInt8}) = GC.@preserve id check(h, md2_comm_init(h.ptr, Int32(rank), Int32(world), hptr(id)), "comm_init")
comm_destroy!(h) = check(h, md2_comm_destroy(h.ptr), "comm_destroy")
function comm_world(h); r = Ref(Int32(0)); w = Ref(Int32(1)); GC.@preserve r w md2_comm_world(h.ptr, Base.unsafe_convert(Ptr{Int32}, r), Base.unsafe_convert(Ptr{Int32}, w)); Int(r[]), Int(w[]); end
# the [lo, hi) slices (0-based float offsets) of the flat gradient buffer md2_backward all-reduces, in completion order; host-only
function comm_bucket_plan(; batch, channels, width, height, encoder_depth=18, scale_levels=collect(2:5))
    cfg = Ref(MD2Config(batch, channels, height, width, false, 0, 0.1, 100.0, 1e-3, 1e-4, encoder_depth, 0, 2, 1, 3, pad5(scale_levels, Int32), pad5(Float64[], Float64)))
    lo = zeros(Int64, 4); hi = zeros(Int64, 4); n = Ref(Int64(0))
    GC.@preserve lo hi n (rc = md2_comm_bucket_plan(cfg, pointer(lo), pointer(hi), Base.unsafe_convert(Ptr{Int64}, n)); rc == 0 || error("md2_comm_bucket_plan failed ($rc)"))
    [(lo[i], hi[i]) for i in 1:4 if hi[i] > lo[i]], n[]
end
allreduce_grads!(h) = check(h, md2_allreduce_grads(h.ptr), "allreduce_grads")
comm_allreduce!(h, x::ROCArray{Float32}) = (check(h, md2_comm_allreduce(h.ptr, dptr(x), length(x)), "comm_allreduce"); x)
broadcast_state!(h, root=0) = check(h, md2_broadcast_state(h.ptr, Int32(root)), "broadcast_state")
function grad_buffer(h); p = Ref{Ptr{Cvoid}}(C_NULL); n = Ref(Int64(0)); GC.@preserve p n check(h, md2_grad_buffer(h.ptr, Base.unsafe_convert(Ptr{Ptr{Cvoid}}, p), Base.unsafe_convert(Ptr{Int64}, n)), "grad_buffer"); p[], n[]; end
function param_buffer(h); p = Ref{Ptr{Cvoid}}(C_NULL); n = Ref(Int64(0)); GC.@preserve p n check(h, md2_param_buffer(h.ptr, Base.unsafe_convert(Ptr{Ptr{Cvoid}}, p), Base.unsafe_convert(Ptr{Int64}, n)), "param_buffer"); p[], n[]; end

# ---- per-class device timing (bench / profiling) ------------------------------------------------------------------------------------------------
profile_enable!(h, on::Bool) = check(h, md2_profile_enable(h.ptr, Int32(on)), "profile_enable")
profile_reset!(h) = check(h, md2_profile_reset(h.ptr), "profile_reset")
function profile_get(h, cls)
    ms = Ref(0.0); n = Ref(Int64(0)); fl = Ref(0.0); by = Ref(0.0)
    GC.@preserve ms n fl by check(h, md2_profile_get(h.ptr, Int32(cls), Base.unsafe_convert(Ptr{Float64}, ms), Base.unsafe_convert(Ptr{Int64}, n), Base.unsafe_convert(Ptr{Float64}, fl), Base.unsafe_convert(Ptr{Float64}, by)), "profile_get")
    ex = Ref(0.0)
    GC.@preserve ex check(h, md2_profile_get_executed(h.ptr, Int32(cls), Base.unsafe_convert(Ptr{Float64}, ex)), "profile_get_executed")
    (ms = ms[], launches = n[], flops = fl[], bytes = by[], flops_executed = ex[])       # executed < algorithmic where the Winograd kernels ran
end
profile_dump(h, path::String) = check(h, md2_profile_dump(h.ptr, path), "profile_dump")

# ---- op-level methods: device dispatch like src/utils.jl:1-11 (AbstractArray vs device array) ------------------------------------------------------
ok(rc, what) = rc == 0 ? nothing : error("libmd2hip $what failed ($rc)")
# (ssim::SSIM)(x, y), src/utils.jl:25
function ssim(x::ROCArray{Float32,4}, y::ROCArray{Float32,4})
    W, H, C, N = size(x); out = similar(x); ok(md2_ssim_fwd(dptr(x), dptr(y), dptr(out), Int32(C * N), Int32(H), Int32(W), NOSTREAM), "ssim_fwd"); out
end
function ∇ssim(x::ROCArray{Float32,4}, y::ROCArray{Float32,4}, Δ::ROCArray{Float32,4})
    W, H, C, N = size(x); dx = similar(x); dy = similar(y)
    ok(md2_ssim_bwd(dptr(x), dptr(y), dptr(Δ), dptr(dx), dptr(dy), Int32(C * N), Int32(H), Int32(W), NOSTREAM), "ssim_bwd"); dx, dy
end
ChainRulesCore.rrule(::typeof(ssim), x::ROCArray{Float32,4}, y::ROCArray{Float32,4}) = ssim(x, y), Δ -> (NoTangent(), ∇ssim(x, y, ROCArray{Float32}(unthunk(Δ)))...)
# photometric_loss(ssim, predicted, target), src/training.jl:1-7
function photometric_loss(pred::ROCArray{Float32,4}, target::ROCArray{Float32,4})
    W, H, C, N = size(pred); out = ROCArray{Float32}(undef, W, H, 1, N)
    ok(md2_photometric_loss_fwd(dptr(pred), dptr(target), dptr(out), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "photometric_loss_fwd"); out
end
function ∇photometric_loss(pred::ROCArray{Float32,4}, target::ROCArray{Float32,4}, Δ::ROCArray{Float32,4})       # wrt `pred` (the target is data)
    W, H, C, N = size(pred); dp = similar(pred)
    ok(md2_photometric_loss_bwd(dptr(pred), dptr(target), dptr(Δ), dptr(dp), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "photometric_loss_bwd"); dp
end
ChainRulesCore.rrule(::typeof(photometric_loss), pred::ROCArray{Float32,4}, target::ROCArray{Float32,4}) =
    photometric_loss(pred, target), Δ -> (NoTangent(), ∇photometric_loss(pred, target, ROCArray{Float32}(unthunk(Δ))), NoTangent())
# prediction_loss(ssim, predictions, target), src/training.jl:13-15; predictions stacked (W,H,C,N,k)
function prediction_loss(preds::ROCArray{Float32,5}, target::ROCArray{Float32,4})
    W, H, C, N, k = size(preds); out = ROCArray{Float32}(undef, W, H, 1, N)
    ok(md2_prediction_loss_fwd(dptr(preds), Int32(k), dptr(target), dptr(out), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "prediction_loss_fwd"); out
end
function ∇prediction_loss(preds::ROCArray{Float32,5}, target::ROCArray{Float32,4}, Δ::ROCArray{Float32,4})         # the gradient of `minimum` goes to the first arg-min
    W, H, C, N, k = size(preds); dp = similar(preds)
    ok(md2_prediction_loss_bwd(dptr(preds), Int32(k), dptr(target), dptr(Δ), dptr(dp), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "prediction_loss_bwd"); dp
end
ChainRulesCore.rrule(::typeof(prediction_loss), preds::ROCArray{Float32,5}, target::ROCArray{Float32,4}) =
    prediction_loss(preds, target), Δ -> (NoTangent(), ∇prediction_loss(preds, target, ROCArray{Float32}(unthunk(Δ))), NoTangent())
# _apply_mask(mask, warp_loss), src/training.jl:17-19
function _apply_mask(mask::ROCArray{Float32,4}, warp_loss::ROCArray{Float32,4})
    size(mask) == size(warp_loss) || error("_apply_mask: sizes differ")
    out = similar(warp_loss); ok(md2_apply_mask(dptr(mask), dptr(warp_loss), dptr(out), length(out), NOSTREAM), "apply_mask"); out
end
# disparity_to_depth, src/utils.jl:175-179
function disparity_to_depth(d::ROCArray{Float32}, min_depth, max_depth)
    out = similar(d); ok(md2_disparity_to_depth(dptr(d), dptr(out), length(d), Float64(min_depth), Float64(max_depth), NOSTREAM), "disparity_to_depth"); out
end
# Backproject(depth (1,P,N), invK), src/utils.jl:63-65 -> (3,P,N)
function backproject(depth::ROCArray{Float32,3}, invK::ROCArray{Float32,2}, W, H)
    N = size(depth, 3); out = ROCArray{Float32}(undef, 3, W * H, N)
    ok(md2_backproject(dptr(depth), dptr(invK), dptr(out), Int32(N), Int32(H), Int32(W), NOSTREAM), "backproject"); out
end
# Project(points, K, R, t), src/utils.jl:95-99 -> (2,P,N)
function project(pts::ROCArray{Float32,3}, K::ROCArray{Float32,2}, R::ROCArray{Float32,3}, t::ROCArray{Float32,3}, W, H)
    N = size(pts, 3); uv = ROCArray{Float32}(undef, 2, W * H, N)
    ok(md2_project(dptr(pts), dptr(K), dptr(R), dptr(t), dptr(uv), Int32(N), Int32(H), Int32(W), NOSTREAM), "project"); uv
end
# NNlib.grid_sample(input, grid; padding_mode), call site src/training.jl:56, and its pullback w.r.t. the grid
function grid_sample(input::ROCArray{Float32,4}, grid::ROCArray{Float32,4}; padding_mode=:zeros)
    W, H, C, N = size(input); _, Wo, Ho, _ = size(grid); out = ROCArray{Float32}(undef, Wo, Ho, C, N)
    ok(md2_grid_sample_fwd(dptr(input), dptr(grid), dptr(out), Int32(N), Int32(C), Int32(H), Int32(W), Int32(Ho), Int32(Wo), Int32(padding_mode == :border), NOSTREAM), "grid_sample_fwd"); out
end
function ∇grid_sample_grid(input::ROCArray{Float32,4}, grid::ROCArray{Float32,4}, Δ::ROCArray{Float32,4}; padding_mode=:zeros)
    W, H, C, N = size(input); _, Wo, Ho, _ = size(grid); dg = similar(grid)
    ok(md2_grid_sample_bwd_grid(dptr(input), dptr(grid), dptr(Δ), dptr(dg), Int32(N), Int32(C), Int32(H), Int32(W), Int32(Ho), Int32(Wo), Int32(padding_mode == :border), NOSTREAM), "grid_sample_bwd_grid"); dg
end
# the fused disparity -> depth -> Backproject -> Project -> grid_sample(:border) of src/training.jl:48-57 (the `warp` src/simple_depth.jl:30 calls)
function warp(disp::ROCArray{Float32,4}, src::ROCArray{Float32,4}, K::ROCArray{Float32,2}, invK::ROCArray{Float32,2}, R::ROCArray{Float32,3}, t::ROCArray{Float32,3}, min_depth, max_depth)
    W, H, C, N = size(src); out = similar(src)
    ok(md2_warp_fwd(dptr(disp), dptr(src), dptr(K), dptr(invK), dptr(R), dptr(t), dptr(out), Int32(N), Int32(C), Int32(H), Int32(W), Float64(min_depth), Float64(max_depth), NOSTREAM), "warp_fwd"); out
end
# its pullback wrt the disparity and the pose (the source image is data): dR (9,N) column-major 3x3 per sample, dt (3,N)
function ∇warp(disp::ROCArray{Float32,4}, src::ROCArray{Float32,4}, K::ROCArray{Float32,2}, invK::ROCArray{Float32,2}, R::ROCArray{Float32,3}, t::ROCArray{Float32,3}, min_depth, max_depth, Δ::ROCArray{Float32,4})
    W, H, C, N = size(src); dd = similar(disp); dR = ROCArray{Float32}(undef, 3, 3, N); dt = similar(t)
    ok(md2_warp_bwd(dptr(disp), dptr(src), dptr(K), dptr(invK), dptr(R), dptr(t), dptr(Δ), dptr(dd), dptr(dR), dptr(dt), Int32(N), Int32(C), Int32(H), Int32(W), Float64(min_depth), Float64(max_depth), NOSTREAM), "warp_bwd"); dd, dR, dt
end
function ChainRulesCore.rrule(::typeof(warp), disp::ROCArray{Float32,4}, src::ROCArray{Float32,4}, K, invK, R::ROCArray{Float32,3}, t::ROCArray{Float32,3}, min_depth, max_depth)
    out = warp(disp, src, K, invK, R, t, min_depth, max_depth)
    function warp_pullback(Δ)
        dd, dR, dt = ∇warp(disp, src, K, invK, R, t, min_depth, max_depth, ROCArray{Float32}(unthunk(Δ)))
        NoTangent(), dd, NoTangent(), NoTangent(), NoTangent(), dR, dt, NoTangent(), NoTangent()
    end
    out, warp_pullback
end
# NNlib.upsample_bilinear(x; size), src/training.jl:45, and its pullback
function upsample_bilinear(x::ROCArray{Float32,4}; size)
    W, H, C, N = Base.size(x); Wo, Ho = size; out = ROCArray{Float32}(undef, Wo, Ho, C, N)
    ok(md2_upsample_bilinear_fwd(dptr(x), dptr(out), Int32(C * N), Int32(H), Int32(W), Int32(Ho), Int32(Wo), NOSTREAM), "upsample_bilinear_fwd"); out
end
function ∇upsample_bilinear(Δ::ROCArray{Float32,4}; size)
    Wo, Ho, C, N = Base.size(Δ); W, H = size; dx = ROCArray{Float32}(undef, W, H, C, N)
    ok(md2_upsample_bilinear_bwd(dptr(Δ), dptr(dx), Int32(C * N), Int32(H), Int32(W), Int32(Ho), Int32(Wo), NOSTREAM), "upsample_bilinear_bwd"); dx
end
# hat / so3_exp_map (+ pullback) / composeT, src/utils.jl:101-141,181-188
function hat(rvec::ROCArray{Float32,2}); N = size(rvec, 2); S = ROCArray{Float32}(undef, 3, 3, N); ok(md2_hat(dptr(rvec), dptr(S), Int32(N), NOSTREAM), "hat"); S; end
function so3_exp_map(rvec::ROCArray{Float32,2}); N = size(rvec, 2); R = ROCArray{Float32}(undef, 3, 3, N); ok(md2_so3_exp_map(dptr(rvec), dptr(R), Int32(N), NOSTREAM), "so3_exp_map"); R; end
function ∇so3_exp_map(rvec::ROCArray{Float32,2}, ΔR::ROCArray{Float32,3}); N = size(rvec, 2); dr = similar(rvec); ok(md2_so3_exp_map_bwd(dptr(rvec), dptr(ΔR), dptr(dr), Int32(N), NOSTREAM), "so3_exp_map_bwd"); dr; end
function composeT(rvec::ROCArray{Float32,2}, t::ROCArray{Float32,3}, invert::Bool)
    N = size(rvec, 2); R = ROCArray{Float32}(undef, 3, 3, N); to = similar(t)
    ok(md2_composeT(dptr(rvec), dptr(t), Int32(invert), dptr(R), dptr(to), Int32(N), NOSTREAM), "composeT"); R, to
end
# smooth_loss(disparity (W,H,N), image (W,H,C,N)), src/utils.jl:159-173 -> device scalar
function smooth_loss(disp::ROCArray{Float32,3}, image::ROCArray{Float32,4})
    W, H, C, N = size(image); out = ROCArray{Float32}(undef, 1)
    ok(md2_smooth_loss(dptr(disp), dptr(image), dptr(out), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "smooth_loss"); out
end
function ∇smooth_loss(disp::ROCArray{Float32,3}, image::ROCArray{Float32,4}, Δ::ROCArray{Float32,1})                 # wrt the disparity (the image is data); Δ: upstream scalar on the device
    W, H, C, N = size(image); dd = similar(disp)
    ok(md2_smooth_loss_bwd(dptr(disp), dptr(image), dptr(Δ), dptr(dd), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "smooth_loss_bwd"); dd
end
ChainRulesCore.rrule(::typeof(smooth_loss), disp::ROCArray{Float32,3}, image::ROCArray{Float32,4}) =
    smooth_loss(disp, image), Δ -> (NoTangent(), ∇smooth_loss(disp, image, ROCArray{Float32}(collect(Float32, unthunk(Δ)))), NoTangent())
# device memory for hosts without an array package: md2_alloc / md2_free (SURVEY.md section 8b ownership rule)
function alloc(bytes::Integer); p = Ref{Ptr{Cvoid}}(C_NULL); GC.@preserve p ok(md2_alloc(Base.unsafe_convert(Ptr{Ptr{Cvoid}}, p), Int64(bytes)), "alloc"); p[]; end
free(p::Ptr{Cvoid}) = ok(md2_free(p), "free")
# Flux Conv / BatchNorm / MaxPool / ADAM as the network uses them (parity tests): weights (kW,kH,Cin,Cout), pad_mode 0 zeros / 1 reflect, act 0 none 1 relu 2 elu 3 σ
function conv2d(x::ROCArray{Float32,4}, w::ROCArray{Float32,4}, b; stride=1, pad=0, pad_mode=0, act=0)
    W, H, Cin, N = size(x); k, _, _, Cout = size(w); pe = pad_mode == 1 ? 1 : pad
    y = ROCArray{Float32}(undef, (W + 2pe - k) ÷ stride + 1, (H + 2pe - k) ÷ stride + 1, Cout, N)
    ok(md2_conv2d_fwd(dptr(x), dptr(w), dptr(b), dptr(y), Int32(N), Int32(Cin), Int32(H), Int32(W), Int32(Cout), Int32(k), Int32(stride), Int32(pad), Int32(pad_mode), Int32(act), NOSTREAM), "conv2d_fwd"); y
end
function ∇conv2d(x::ROCArray{Float32,4}, w::ROCArray{Float32,4}, Δ::ROCArray{Float32,4}; stride=1, pad=0, pad_mode=0)
    W, H, Cin, N = size(x); k, _, _, Cout = size(w); dx = similar(x); dw = similar(w); db = ROCArray{Float32}(undef, Cout)
    ok(md2_conv2d_bwd(dptr(x), dptr(w), dptr(Δ), dptr(dx), dptr(dw), dptr(db), Int32(N), Int32(Cin), Int32(H), Int32(W), Int32(Cout), Int32(k), Int32(stride), Int32(pad), Int32(pad_mode), NOSTREAM), "conv2d_bwd"); dx, dw, db
end
function conv2d_bnstats(x::ROCArray{Float32,4}, w::ROCArray{Float32,4}; stride=1, pad=0)          # encoder conv + the BatchNorm statistics of its epilogue
    W, H, Cin, N = size(x); k, _, _, Cout = size(w)
    y = ROCArray{Float32}(undef, (W + 2pad - k) ÷ stride + 1, (H + 2pad - k) ÷ stride + 1, Cout, N); mean = ROCArray{Float32}(undef, Cout); invstd = similar(mean)
    ok(md2_conv2d_bnstats_fwd(dptr(x), dptr(w), dptr(y), dptr(mean), dptr(invstd), Int32(N), Int32(Cin), Int32(H), Int32(W), Int32(Cout), Int32(k), Int32(stride), Int32(pad), NOSTREAM), "conv2d_bnstats_fwd"); y, mean, invstd
end
function stem_conv(x::ROCArray{Float32,4}, w::ROCArray{Float32,4})                                     # ResNet stem 7x7 / stride 2 / pad 3, direct kernel
    W, H, C, N = size(x); z = ROCArray{Float32}(undef, W ÷ 2, H ÷ 2, 64, N); mean = ROCArray{Float32}(undef, 64); invstd = similar(mean)
    ok(md2_stem_conv_fwd(dptr(x), dptr(w), dptr(z), dptr(mean), dptr(invstd), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "stem_conv_fwd"); z, mean, invstd
end
function ∇stem_conv_weight(x::ROCArray{Float32,4}, Δ::ROCArray{Float32,4})
    W, H, C, N = size(x); dw = ROCArray{Float32}(undef, 7, 7, C, 64)
    ok(md2_stem_conv_wgrad(dptr(x), dptr(Δ), dptr(dw), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "stem_conv_wgrad"); dw
end
function head_conv(x::ROCArray{Float32,4}, w::ROCArray{Float32,4}, b::ROCArray{Float32,1})             # DecoderBlock(Conv((3,3), c => 1, σ)), src/depth_decoder.jl:43
    W, H, Cin, N = size(x); y = ROCArray{Float32}(undef, W, H, 1, N)
    ok(md2_head_conv_fwd(dptr(x), dptr(w), dptr(b), dptr(y), Int32(N), Int32(Cin), Int32(H), Int32(W), NOSTREAM), "head_conv_fwd"); y
end
function ∇head_conv(x, w, y, Δ; x_act=0)
    W, H, Cin, N = size(x); dx = similar(x); dw = similar(w); db = ROCArray{Float32}(undef, 1)
    ok(md2_head_conv_bwd(dptr(x), dptr(w), dptr(y), dptr(Δ), dptr(dx), dptr(dw), dptr(db), Int32(N), Int32(Cin), Int32(H), Int32(W), Int32(x_act), NOSTREAM), "head_conv_bwd"); dx, dw, db
end
function batchnorm_train(x::ROCArray{Float32,4}, γ, β; relu=false)
    W, H, C, N = size(x); y = similar(x); ok(md2_bn_train_fwd(dptr(x), dptr(γ), dptr(β), dptr(y), Int32(N), Int32(C), Int32(H), Int32(W), Int32(relu), NOSTREAM), "bn_train_fwd"); y
end
function ∇batchnorm_train(x::ROCArray{Float32,4}, γ, Δ)
    W, H, C, N = size(x); dx = similar(x); dγ = similar(γ); dβ = similar(γ)
    ok(md2_bn_train_bwd(dptr(x), dptr(γ), dptr(Δ), dptr(dx), dptr(dγ), dptr(dβ), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "bn_train_bwd"); dx, dγ, dβ
end
function maxpool3s2(x::ROCArray{Float32,4})
    W, H, C, N = size(x); y = ROCArray{Float32}(undef, (W - 1) ÷ 2 + 1, (H - 1) ÷ 2 + 1, C, N); ok(md2_maxpool3s2_fwd(dptr(x), dptr(y), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "maxpool3s2_fwd"); y
end
function ∇maxpool3s2(x::ROCArray{Float32,4}, Δ)
    W, H, C, N = size(x); dx = similar(x); ok(md2_maxpool3s2_bwd(dptr(x), dptr(Δ), dptr(dx), Int32(N), Int32(C), Int32(H), Int32(W), NOSTREAM), "maxpool3s2_bwd"); dx
end
adam_raw!(p, g, m, v, step; lr=1e-4, grad_scale=1f0) = ok(md2_adam_raw(dptr(p), dptr(g), dptr(m), dptr(v), length(p), Int32(step), Float64(lr), Float32(grad_scale), NOSTREAM), "adam_raw")

end # module
