# Monodepth2HIP.jl — the Julia `ccall` shim over libmd2hip.so (include/md2hip.h).
#
# NOT EXECUTED IN THIS REPO'S CI: the build image has no Julia.  It is mechanically derived from the C header and is what a
# maintainer of pxl-th/Monodepth2.jl adds to keep the module's call surface (src/Monodepth.jl, model.jl, training.jl,
# utils.jl) while running the training step on an MI355X.  Device memory is owned by AMDGPU.jl `ROCArray{Float32}`s
# (any array type exposing a device pointer works); Julia's column-major (W,H,C,N) bytes are passed unchanged.
module Monodepth2HIP

using AMDGPU

const LIB = get(ENV, "MD2HIP_LIB", joinpath(@__DIR__, "..", "libmd2hip.so"))

struct MD2Config
    batch::Int32; channels::Int32; height::Int32; width::Int32
    automasking::Int32; device::Int32
    min_depth::Float64; max_depth::Float64; disparity_smoothness::Float64; lr::Float64
    encoder_depth::Int32; precision::Int32       # ResidualNetwork(18 | 50); 0 = f32, 1 = bf16 MFMA operands
end

mutable struct Handle
    ptr::Ptr{Cvoid}
end

function check(h, rc, what)
    rc == 0 && return
    msg = unsafe_string(ccall((:md2_last_error, LIB), Cstring, (Ptr{Cvoid},), h === nothing ? C_NULL : h.ptr))
    error("libmd2hip $what failed ($rc): $msg")          # reference convention: plain Julia exceptions
end

dptr(x::ROCArray{Float32}) = Ptr{Cvoid}(UInt(pointer(x)))
dptr(::Nothing) = C_NULL

# Params (src/Monodepth.jl:32-42) -> handle
function Handle(; batch, channels, width, height, automasking=false, device=0,
                min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, lr=1e-4, encoder_depth=18, precision=0)
    cfg = Ref(MD2Config(batch, channels, height, width, automasking, device, min_depth, max_depth, disparity_smoothness, lr, encoder_depth, precision))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:md2_create, LIB), Int32, (Ref{MD2Config}, Ref{Ptr{Cvoid}}), cfg, out)
    check(nothing, rc, "md2_create")
    h = Handle(out[]); finalizer(x -> ccall((:md2_destroy, LIB), Int32, (Ptr{Cvoid},), x.ptr), h); h
end

# TrainCache.K / invK (src/Monodepth.jl:113): 3x3 column-major
set_intrinsics!(h, K::Matrix{Float32}, invK::Matrix{Float32}) =
    check(h, ccall((:md2_set_intrinsics, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), h.ptr, K, invK), "set_intrinsics")

# Flux.params(model) as one flat vector (order: DESIGN.md §3; conv weights (kW,kH,Cin,Cout))
params_count(h) = ccall((:md2_params_count, LIB), Int64, (Ptr{Cvoid},), h.ptr)
load_params!(h, θ::Vector{Float32}) = check(h, ccall((:md2_params_set, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), h.ptr, θ, length(θ)), "params_set")
function params(h); θ = Vector{Float32}(undef, params_count(h)); check(h, ccall((:md2_params_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), h.ptr, θ, length(θ)), "params_get"); θ; end
function grads(h); g = Vector{Float32}(undef, params_count(h)); check(h, ccall((:md2_grads_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), h.ptr, g, length(g)), "grads_get"); g; end

# (m::Model)(x, source_ids, target_id), src/model.jl:8-20.  x: ROCArray (W,H,C,3,N)
function model_forward(h, x::ROCArray{Float32,5})
    W, H, C, L, N = size(x)
    disps = [ROCArray{Float32}(undef, W >> s, H >> s, 1, N) for s in (3, 2, 1, 0)]
    rvec = ROCArray{Float32}(undef, 3, N, 2); tvec = ROCArray{Float32}(undef, 3, N, 2)       # Julia (3,N,2) == C [2][N][3]
    check(h, ccall((:md2_model_forward, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                h.ptr, dptr(x), dptr(disps[1]), dptr(disps[2]), dptr(disps[3]), dptr(disps[4]), dptr(rvec), dptr(tvec)), "model_forward")
    disps, rvec, tvec
end

# train_loss(model, x, auto_loss, cache, parameters, false)[1], src/training.jl:21-78 -> device scalar
function train_loss(h, x::ROCArray{Float32,5}, auto_loss)
    loss = ROCArray{Float32}(undef, 1)
    check(h, ccall((:md2_train_loss, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, dptr(x), dptr(auto_loss), dptr(loss)), "train_loss")
    loss
end
# gradient(θ) do train_loss(...) end  — as a ChainRulesCore.rrule the pullback is `Δ -> (md2_backward; grads(h) .* Δ)`
backward!(h) = check(h, ccall((:md2_backward, LIB), Int32, (Ptr{Cvoid},), h.ptr), "backward")
# Flux.Optimise.update!(ADAM(1e-4), θ, ∇), src/Monodepth.jl:170
adam_step!(h, grad_scale=1f0) = check(h, ccall((:md2_adam_step, LIB), Int32, (Ptr{Cvoid}, Float32), h.ptr, grad_scale), "adam_step")
# the loop body src/Monodepth.jl:156-176 in one call
function train_step!(h, x, auto_loss=nothing; update=true)
    loss = ROCArray{Float32}(undef, 1)
    check(h, ccall((:md2_train_step, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32), h.ptr, dptr(x), dptr(auto_loss), dptr(loss), update), "train_step")
    loss
end
# automasking_loss(ssim, x, x[:,:,:,target,:]; source_ids), src/training.jl:9-11
function automasking_loss(h, x::ROCArray{Float32,5})
    W, H, _, _, N = size(x); out = ROCArray{Float32}(undef, W, H, 1, N)
    check(h, ccall((:md2_automasking_loss, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, dptr(x), dptr(out)), "automasking_loss"); out
end

# eval_disparity(m, x), src/model.jl:28.  x: ROCArray (W,H,C,n) single frames, n <= batch
function eval_disparity(h, x::ROCArray{Float32,4})
    W, H, C, n = size(x)
    disps = [ROCArray{Float32}(undef, W >> s, H >> s, 1, n) for s in (3, 2, 1, 0)]
    check(h, ccall((:md2_eval_disparity, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                h.ptr, dptr(x), n, dptr(disps[1]), dptr(disps[2]), dptr(disps[3]), dptr(disps[4])), "eval_disparity")
    disps
end
# slow_depth's objective + pullbacks (src/simple_depth.jl:25-42): disp (1,W,H,N) == [N][1][H][W]
function simple_depth_loss(h, x, disp, rvec, tvec)
    loss = ROCArray{Float32}(undef, 1); dd = similar(disp); dr = similar(rvec); dt = similar(tvec)
    check(h, ccall((:md2_simple_depth_loss, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                h.ptr, dptr(x), dptr(disp), dptr(rvec), dptr(tvec), dptr(loss), dptr(dd), dptr(dr), dptr(dt)), "simple_depth_loss")
    loss, dd, dr, dt
end

# ---- op-level methods: device dispatch like src/utils.jl:1-11 (AbstractArray vs device array) --------------------------------
# (ssim::SSIM)(x::ROCArray, y::ROCArray), src/utils.jl:25
function ssim(x::ROCArray{Float32,4}, y::ROCArray{Float32,4})
    W, H, C, N = size(x); out = similar(x)
    rc = ccall((:md2_ssim_fwd, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}), dptr(x), dptr(y), dptr(out), C * N, H, W, C_NULL)
    check(nothing, rc, "ssim_fwd"); out
end
# so3_exp_map(rvec::ROCArray), src/utils.jl:102
function so3_exp_map(rvec::ROCArray{Float32,2})
    N = size(rvec, 2); R = ROCArray{Float32}(undef, 3, 3, N)
    check(nothing, ccall((:md2_so3_exp_map, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}), dptr(rvec), dptr(R), N, C_NULL), "so3_exp_map"); R
end
# NNlib.grid_sample(input::ROCArray, grid::ROCArray; padding_mode), call site src/training.jl:56
function grid_sample(input::ROCArray{Float32,4}, grid::ROCArray{Float32,4}; padding_mode=:zeros)
    W, H, C, N = size(input); _, Wo, Ho, _ = size(grid); out = ROCArray{Float32}(undef, Wo, Ho, C, N)
    rc = ccall((:md2_grid_sample_fwd, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Int32, Int32, Ptr{Cvoid}),
               dptr(input), dptr(grid), dptr(out), N, C, H, W, Ho, Wo, padding_mode == :border, C_NULL)
    check(nothing, rc, "grid_sample_fwd"); out
end
# ... the remaining op-level entry points (md2_backproject, md2_project, md2_composeT, md2_smooth_loss, md2_disparity_to_depth,
# md2_upsample_bilinear_*, md2_conv2d_*, md2_bn_train_*, md2_maxpool3s2_*) bind the same way: see include/md2hip.h.

end # module
