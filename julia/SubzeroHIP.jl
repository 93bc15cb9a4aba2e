# SubzeroHIP.jl -- Subzero.jl on an MI355X: the three per-timestep hot calls of `timestep_sim!`
# (src/simulation_components/simulation.jl:94-220) routed to libsubzero_hip.so through its C-ABI
# (include/subzero_hip.h).  Nothing in Subzero.jl is edited:
#
#     using Subzero, SubzeroHIP
#     sim = Simulation(...)                       # Float64 simulation, as usual
#     eng = SubzeroHIP.enable!(sim; device = 0)   # from here on the engine does the hot path
#     run!(sim)                                   # unchanged driver loop, unchanged output writers
#     SubzeroHIP.disable!()                       # back to the CPU path (and frees the device state)
#
# How it hooks in.  `timestep_sim!` calls
#     timestep_collisions!(floes, n_init_floes, domain, consts, Δt, collision_settings, spinlock)   collisions.jl:734
#     timestep_coupling!(model, Δt, consts, coupling_settings, floe_settings)                      coupling.jl:1705
#     timestep_floe_properties!(floes, tstep, Δt, floe_settings)                                   update_floe.jl:469
# whose reference methods leave the settings / constants arguments untyped.  This module adds methods that are MORE
# SPECIFIC in exactly those arguments (Float64 floes, `Constants{Float64}`, `CollisionSettings{Float64}`,
# `FloeSettings`), so Julia's dispatch picks them for every Float64 simulation once the module is loaded; each of them
# looks at `ENGINE[]` and either runs on the device or `invoke`s the reference method it shadows (no engine enabled:
# bit-for-bit the CPU behaviour).  The call sites in `timestep_sim!` do not change.
#
# Two ways to run:
#   * process mode (the three methods below): upload the floe columns, one device call, download what the reference
#     function would have mutated.  Works with every other process (fracture, ridging, welding, simplification, all
#     output writers), which keep seeing ordinary `StructArray{Floe}` state.
#   * resident mode (`run_resident!`): when no host-side process needs the floes between steps, whole batches of
#     `timestep_sim!` run with the state resident in HBM (`sz_step`); the batch ends early when a floe is tagged
#     remove / fuse, so that `simplify_floes!` runs at the step the reference would run it.
#
# NOT RUN in the build container (it has no Julia): the struct mirrors below are checked against the C header by
# tests/test_host_cpu.py::test_julia_struct_mirrors_match_the_header (field order, sizes and offsets), and the same
# call sequences are exercised on the GPU by the Python mirror of this file (subzero.jl_amd/host.py).
module SubzeroHIP

using Subzero
using StructArrays
import GeometryOps.GeoInterface as GI

export enable!, disable!, run_resident!, HIPEngine

const lib = get(ENV, "SUBZERO_HIP_LIB", "libsubzero_hip.so")

# ------------------------------------------------------------------------------------------------ C-ABI mirrors
# include/subzero_hip.h: sz_params
struct SzParams
    E::Float64
    nu::Float64
    mu::Float64
    rho_o::Float64
    rho_a::Float64
    Cd_io::Float64
    Cd_ia::Float64
    f::Float64
    turn_theta::Float64
    floe_floe_max_overlap::Float64
    floe_domain_max_overlap::Float64
    rho_i::Float64
    max_floe_height::Float64
    maximum_xi::Float64
    lambda::Float64
    coupling_dd::Int32
    _pad::Int32
end

# include/subzero_hip.h: sz_floe_columns (host pointers; C_NULL = column absent)
mutable struct SzFloeColumns
    cx::Ptr{Float64}
    cy::Ptr{Float64}
    rmax::Ptr{Float64}
    area::Ptr{Float64}
    height::Ptr{Float64}
    mass::Ptr{Float64}
    moment::Ptr{Float64}
    alpha::Ptr{Float64}
    u::Ptr{Float64}
    v::Ptr{Float64}
    xi::Ptr{Float64}
    p_dxdt::Ptr{Float64}
    p_dydt::Ptr{Float64}
    p_dalphadt::Ptr{Float64}
    p_dudt::Ptr{Float64}
    p_dvdt::Ptr{Float64}
    p_dxidt::Ptr{Float64}
    fxOA::Ptr{Float64}
    fyOA::Ptr{Float64}
    trqOA::Ptr{Float64}
    hflx_factor::Ptr{Float64}
    overarea::Ptr{Float64}
    coll_fx::Ptr{Float64}
    coll_fy::Ptr{Float64}
    coll_trq::Ptr{Float64}
    stress_accum::Ptr{Float64}
    stress_instant::Ptr{Float64}
    strain::Ptr{Float64}
    id::Ptr{Int64}
    ghost_id::Ptr{Int64}
    status::Ptr{Int32}
    vert_off::Ptr{Int32}
    vx::Ptr{Float64}
    vy::Ptr{Float64}
    sub_off::Ptr{Int32}
    sx::Ptr{Float64}
    sy::Ptr{Float64}
    ghost_off::Ptr{Int32}
    ghost_idx::Ptr{Int32}
end
SzFloeColumns() = SzFloeColumns(ntuple(_ -> C_NULL, 39)...)

# include/subzero_hip.h: sz_floe_columns_f32 -- the same columns of a Floe{Float32} host (the engine widens them on the way in and rounds
# them on the way out; see the Float32 section at the end of this file)
mutable struct SzFloeColumnsF32
    cx::Ptr{Float32}
    cy::Ptr{Float32}
    rmax::Ptr{Float32}
    area::Ptr{Float32}
    height::Ptr{Float32}
    mass::Ptr{Float32}
    moment::Ptr{Float32}
    alpha::Ptr{Float32}
    u::Ptr{Float32}
    v::Ptr{Float32}
    xi::Ptr{Float32}
    p_dxdt::Ptr{Float32}
    p_dydt::Ptr{Float32}
    p_dalphadt::Ptr{Float32}
    p_dudt::Ptr{Float32}
    p_dvdt::Ptr{Float32}
    p_dxidt::Ptr{Float32}
    fxOA::Ptr{Float32}
    fyOA::Ptr{Float32}
    trqOA::Ptr{Float32}
    hflx_factor::Ptr{Float32}
    overarea::Ptr{Float32}
    coll_fx::Ptr{Float32}
    coll_fy::Ptr{Float32}
    coll_trq::Ptr{Float32}
    stress_accum::Ptr{Float32}
    stress_instant::Ptr{Float32}
    strain::Ptr{Float32}
    id::Ptr{Int64}
    ghost_id::Ptr{Int64}
    status::Ptr{Int32}
    vert_off::Ptr{Int32}
    vx::Ptr{Float32}
    vy::Ptr{Float32}
    sub_off::Ptr{Int32}
    sx::Ptr{Float32}
    sy::Ptr{Float32}
    ghost_off::Ptr{Int32}
    ghost_idx::Ptr{Int32}
end
SzFloeColumnsF32() = SzFloeColumnsF32(ntuple(_ -> C_NULL, 39)...)

# include/subzero_hip.h: sz_stats
struct SzStats
    M::Int64
    N::Int64
    n_ring_points::Int64
    n_sub_points::Int64
    n_pairs::Int64
    n_pair_ring_points::Int64
    n_pair_rows::Int64
    n_elem_items::Int64
    n_elem_rows::Int64
    n_inter_rows::Int64
    n_ghosts::Int64
    warn_height::Int64
    warn_force::Int64
    warn_vel::Int64
    warn_xi::Int64
    n_trace_fail::Int64
    n_halo::Int64
    n_pairs_clipped::Int64
    n_status_remove::Int64
    n_status_fuse::Int64
    n_retry::Int64
    acc_narrow_launches::Int64
    acc_pair_items::Int64
    acc_pair_ring_points::Int64
    acc_pair_rows::Int64
    acc_elem_items::Int64
    acc_elem_rows::Int64
    acc_dir_checks::Int64
    acc_dir_checks_certified::Int64
end

const SZ_COLLISIONS_ON = Int32(1)
const SZ_COUPLING_ON = Int32(2)
const SZ_NO_STOP = Int32(4)
boundary_kind(::OpenBoundary) = Int32(0)
boundary_kind(::PeriodicBoundary) = Int32(1)
boundary_kind(::CollisionBoundary) = Int32(2)
boundary_kind(::MovingBoundary) = Int32(3)

# ------------------------------------------------------------------------------------------------ the engine
mutable struct HIPEngine
    ctx::Ptr{Cvoid}
    two_way::Bool
    Nx::Int
    Ny::Int
    max_vertices::Int32          # SimplificationSettings / FloeSettings values of the simulation the engine was made for
    min_floe_area::Float64       # (sz_simplify_check)
    min_floe_height::Float64
end

const ENGINE = Ref{Union{Nothing, HIPEngine}}(nothing)

last_error(eng) = unsafe_string(@ccall lib.sz_last_error(eng.ctx::Ptr{Cvoid})::Cstring)
function check(eng::HIPEngine, rc::Integer)
    rc == 0 || error("libsubzero_hip: error $rc: $(last_error(eng))")
    return
end

# lattice fields are (Nx+1) x (Ny+1) column-major in Julia; the library wants element [ix][iy] at ix*(Ny+1)+iy
lattice(m::AbstractMatrix) = collect(Float64, permutedims(m))
unlattice(buf::Vector{Float64}, Nx, Ny) = permutedims(reshape(buf, Ny + 1, Nx + 1))

"""
    enable!(sim; device = 0) -> HIPEngine

Create the device context for `sim` (constants, settings, domain, topography, ocean / atmosphere lattices) and route
the hot calls of every Float64 simulation to it until `disable!()`.
"""
function enable!(sim; device::Integer = 0)
    disable!()
    ctx = @ccall lib.sz_create(device::Cint)::Ptr{Cvoid}
    ctx == C_NULL && error("sz_create: no HIP device (the engine has no CPU fallback)")
    model, c, cs, fs, cp = sim.model, sim.consts, sim.collision_settings, sim.floe_settings, sim.coupling_settings
    grid = model.grid
    eng = HIPEngine(ctx, cp.two_way_coupling_on, grid.Nx, grid.Ny, Int32(sim.simp_settings.max_vertices),
                    Float64(fs.min_floe_area), Float64(fs.min_floe_height))
    λ = hasproperty(fs.stress_calculator, :λ) ? Float64(fs.stress_calculator.λ) :
        error("SubzeroHIP implements DecayAreaScaledCalculator (stress_calculators.jl:82) only")
    p = Ref(SzParams(c.E, c.ν, c.μ, c.ρo, c.ρa, c.Cd_io, c.Cd_ia, c.f, c.turnθ, cs.floe_floe_max_overlap,
                     cs.floe_domain_max_overlap, fs.ρi, fs.max_floe_height, fs.maximum_ξ, λ, Int32(cp.Δd), Int32(0)))
    check(eng, @ccall lib.sz_set_params(ctx::Ptr{Cvoid}, p::Ptr{SzParams})::Cint)
    push_domain!(eng, model.domain)
    push_fields!(eng, model)
    if cp.two_way_coupling_on
        check(eng, @ccall lib.sz_set_two_way(ctx::Ptr{Cvoid}, 1::Int32, c.Cd_ao::Float64, c.k::Float64, c.L::Float64,
                                             sim.Δt::Int32)::Cint)
        to, ta = lattice(model.ocean.temp), lattice(model.atmos.temp)
        check(eng, @ccall lib.sz_set_temps(ctx::Ptr{Cvoid}, to::Ptr{Float64}, ta::Ptr{Float64})::Cint)
    end
    ENGINE[] = eng
    return eng
end

function disable!()
    eng = ENGINE[]
    if eng !== nothing
        @ccall lib.sz_destroy(eng.ctx::Ptr{Cvoid})::Cvoid
        ENGINE[] = nothing
    end
    return
end

function push_domain!(eng::HIPEngine, domain)
    bnds = (domain.north, domain.south, domain.east, domain.west)
    kinds = Int32[boundary_kind(b) for b in bnds]
    vals = Float64[b.val for b in bnds]
    rects = Float64[]
    for b in bnds
        (x0, xf), (y0, yf) = GI.extent(b.poly)
        append!(rects, (x0, xf, y0, yf))
    end
    bu = Float64[b isa MovingBoundary ? b.u : 0.0 for b in bnds]
    bv = Float64[b isa MovingBoundary ? b.v : 0.0 for b in bnds]
    check(eng, @ccall lib.sz_set_domain(eng.ctx::Ptr{Cvoid}, kinds::Ptr{Int32}, vals::Ptr{Float64}, rects::Ptr{Float64},
                                        bu::Ptr{Float64}, bv::Ptr{Float64})::Cint)
    topo = domain.topography
    nt = length(topo)
    off = Int32[0]; tx = Float64[]; ty = Float64[]
    for i in 1:nt
        for pt in GI.getpoint(GI.getexterior(topo.poly[i]))
            push!(tx, GI.x(pt)); push!(ty, GI.y(pt))
        end
        push!(off, Int32(length(tx)))
    end
    tcx = Float64[topo.centroid[i][1] for i in 1:nt]; tcy = Float64[topo.centroid[i][2] for i in 1:nt]
    trm = Float64[topo.rmax[i] for i in 1:nt]
    check(eng, @ccall lib.sz_set_topography(eng.ctx::Ptr{Cvoid}, nt::Int32, off::Ptr{Int32}, tx::Ptr{Float64}, ty::Ptr{Float64},
                                            tcx::Ptr{Float64}, tcy::Ptr{Float64}, trm::Ptr{Float64})::Cint)
    return
end

function push_fields!(eng::HIPEngine, model)
    g, o, a = model.grid, model.ocean, model.atmos
    uo, vo, hf, ua, va = lattice(o.u), lattice(o.v), lattice(o.hflx_factor), lattice(a.u), lattice(a.v)
    check(eng, @ccall lib.sz_set_fields(eng.ctx::Ptr{Cvoid}, g.Nx::Int32, g.Ny::Int32, g.x0::Float64, g.xf::Float64,
                                        g.y0::Float64, g.yf::Float64, uo::Ptr{Float64}, vo::Ptr{Float64}, hf::Ptr{Float64},
                                        ua::Ptr{Float64}, va::Ptr{Float64})::Cint)
    return
end

# ------------------------------------------------------------------------------------------------ pack / unpack
# Everything the pointers of an SzFloeColumns point into (kept alive by the caller with GC.@preserve).
struct Packed
    cols::SzFloeColumns
    cx::Vector{Float64}; cy::Vector{Float64}
    coll_fx::Vector{Float64}; coll_fy::Vector{Float64}
    sa::Vector{Float64}; si::Vector{Float64}; strain::Vector{Float64}
    status::Vector{Int32}
    vert_off::Vector{Int32}; vx::Vector{Float64}; vy::Vector{Float64}
    sub_off::Vector{Int32}; sx::Vector{Float64}; sy::Vector{Float64}
    ghost_off::Vector{Int32}; ghost_idx::Vector{Int32}
    id::Vector{Int64}; ghost_id::Vector{Int64}
end

tensor4(m) = (m[1, 1], m[1, 2], m[2, 1], m[2, 2])     # 2x2 -> the library's order 11, 12, 21, 22

"""
    pack(floes, n_parents) -> Packed

The hot columns of `StructArray{Floe{Float64}}` (floe.jl:24-77) as `sz_floe_columns`: scalar columns are passed where
they lie (they are contiguous `Vector{Float64}` / `Vector{Int}` already), ragged ones are flattened to CSR.
"""
function pack(floes::StructArray{<:Floe{Float64}}, n_parents::Integer)
    M = length(floes)
    cx = Float64[c[1] for c in floes.centroid]; cy = Float64[c[2] for c in floes.centroid]
    coll_fx = Float64[f[1, 1] for f in floes.collision_force]; coll_fy = Float64[f[1, 2] for f in floes.collision_force]
    sa = Vector{Float64}(undef, 4M); si = similar(sa); st = similar(sa)
    for i in 1:M
        sa[4i-3:4i] .= tensor4(floes.stress_accum[i]); si[4i-3:4i] .= tensor4(floes.stress_instant[i])
        st[4i-3:4i] .= tensor4(floes.strain[i])
    end
    status = Int32[Int32(s.tag) for s in floes.status]            # active = 1, remove = 2, fuse = 3 (floe.jl:8-12)
    vert_off = Vector{Int32}(undef, M + 1); vert_off[1] = 0
    vx = Float64[]; vy = Float64[]
    for i in 1:M                                                   # the exterior ring, closed (floe_utils.jl:10-17)
        for pt in GI.getpoint(GI.getexterior(floes.poly[i]))
            push!(vx, GI.x(pt)); push!(vy, GI.y(pt))
        end
        vert_off[i+1] = length(vx)
    end
    sub_off = Vector{Int32}(undef, n_parents + 1); sub_off[1] = 0
    sx = Float64[]; sy = Float64[]
    for i in 1:n_parents
        append!(sx, floes.x_subfloe_points[i]); append!(sy, floes.y_subfloe_points[i])
        sub_off[i+1] = length(sx)
    end
    ghost_off = Vector{Int32}(undef, M + 1); ghost_off[1] = 0
    ghost_idx = Int32[]
    for i in 1:M
        append!(ghost_idx, Int32.(floes.ghosts[i] .- 1))           # 0-based in the C-ABI
        ghost_off[i+1] = length(ghost_idx)
    end
    id = Vector{Int64}(floes.id); ghost_id = Vector{Int64}(floes.ghost_id)
    c = SzFloeColumns()
    c.cx = pointer(cx); c.cy = pointer(cy)
    c.rmax = pointer(floes.rmax); c.area = pointer(floes.area); c.height = pointer(floes.height)
    c.mass = pointer(floes.mass); c.moment = pointer(floes.moment); c.alpha = pointer(floes.α)
    c.u = pointer(floes.u); c.v = pointer(floes.v); c.xi = pointer(floes.ξ)
    c.p_dxdt = pointer(floes.p_dxdt); c.p_dydt = pointer(floes.p_dydt); c.p_dalphadt = pointer(floes.p_dαdt)
    c.p_dudt = pointer(floes.p_dudt); c.p_dvdt = pointer(floes.p_dvdt); c.p_dxidt = pointer(floes.p_dξdt)
    c.fxOA = pointer(floes.fxOA); c.fyOA = pointer(floes.fyOA); c.trqOA = pointer(floes.trqOA)
    c.hflx_factor = pointer(floes.hflx_factor); c.overarea = pointer(floes.overarea)
    c.coll_fx = pointer(coll_fx); c.coll_fy = pointer(coll_fy); c.coll_trq = pointer(floes.collision_trq)
    c.stress_accum = pointer(sa); c.stress_instant = pointer(si); c.strain = pointer(st)
    c.id = pointer(id); c.ghost_id = pointer(ghost_id); c.status = pointer(status)
    c.vert_off = pointer(vert_off); c.vx = pointer(vx); c.vy = pointer(vy)
    c.sub_off = pointer(sub_off); c.sx = pointer(sx); c.sy = pointer(sy)
    if M > n_parents
        c.ghost_off = pointer(ghost_off); c.ghost_idx = pointer(ghost_idx)
    end
    return Packed(c, cx, cy, coll_fx, coll_fy, sa, si, st, status, vert_off, vx, vy, sub_off, sx, sy, ghost_off, ghost_idx,
                  id, ghost_id)
end

function upload!(eng::HIPEngine, floes, n_parents)
    P = pack(floes, n_parents)
    GC.@preserve P floes begin
        check(eng, @ccall lib.sz_upload_floes(eng.ctx::Ptr{Cvoid}, length(floes)::Int64, n_parents::Int64,
                                              P.cols::Ref{SzFloeColumns})::Cint)
    end
    return P
end

# floe.interactions of every floe -> the device (calc_stress!, update_floe.jl:392-414, reads them)
function upload_interactions!(eng::HIPEngine, floes)
    M = length(floes)
    off = Vector{Int32}(undef, M + 1); off[1] = 0
    for i in 1:M
        off[i+1] = off[i] + floes.num_inters[i]
    end
    rows = Matrix{Float64}(undef, 7, max(Int(off[end]), 1))          # row-major k x 7 on the C side
    for i in 1:M, k in 1:floes.num_inters[i]
        rows[:, off[i]+k] .= @view floes.interactions[i][k, :]
    end
    check(eng, @ccall lib.sz_upload_interactions(eng.ctx::Ptr{Cvoid}, off::Ptr{Int32}, rows::Ptr{Float64})::Cint)
    return
end

function stats(eng::HIPEngine)
    st = Ref{SzStats}()
    check(eng, @ccall lib.sz_get_stats(eng.ctx::Ptr{Cvoid}, st::Ptr{SzStats})::Cint)
    return st[]
end

# scalar columns come back where they lie; `P` holds the flattened ones
function download!(eng::HIPEngine, floes, P::Packed)
    GC.@preserve P floes begin
        check(eng, @ccall lib.sz_download_floes(eng.ctx::Ptr{Cvoid}, P.cols::Ref{SzFloeColumns})::Cint)
    end
    return
end

function unpack_status!(floes, P::Packed)
    for i in eachindex(floes)
        floes.status[i].tag = Subzero.StatusTag(P.status[i])
    end
end

# ------------------------------------------------------------------------------------------------ the three hot calls
# timestep_collisions! (collisions.jl:734-864): what it mutates is reproduced on the host -- interactions (k x 7, floeidx
# column as the reference stores it), num_inters, overarea, status (+ fuse_idx), collision_force / collision_trq of the
# parents, MovingBoundary walls.
function Subzero.timestep_collisions!(floes::StructArray{<:Floe{Float64}}, n_init_floes, domain::Domain,
                                      consts::Constants{Float64}, Δt, collision_settings::CollisionSettings{Float64}, spinlock)
    eng = ENGINE[]
    if eng === nothing
        return invoke(Subzero.timestep_collisions!, Tuple{StructArray{<:Floe{Float64}}, Any, Any, Any, Any, Any, Any},
                      floes, n_init_floes, domain, consts, Δt, collision_settings, spinlock)
    end
    M = length(floes)
    P = upload!(eng, floes, n_init_floes)           # Julia's add_ghosts! has run: ghost rows and links go up as they are
    check(eng, @ccall lib.sz_timestep_collisions(eng.ctx::Ptr{Cvoid}, n_init_floes::Int64, Δt::Int32)::Cint)
    st = stats(eng)
    off = Vector{Int32}(undef, M + 1); rows = Matrix{Float64}(undef, 7, max(Int(st.n_inter_rows), 1))
    check(eng, @ccall lib.sz_download_interactions(eng.ctx::Ptr{Cvoid}, off::Ptr{Int32}, rows::Ptr{Float64})::Cint)
    download!(eng, floes, P)                        # overarea, collision_trq in place; coll_fx / coll_fy / status in P
    for i in 1:M
        r = off[i]+1:off[i+1]
        floes.interactions[i] = permutedims(rows[:, r])            # k x 7: floeidx xforce yforce xpoint ypoint torque overlap
        floes.num_inters[i] = length(r)
        floes.collision_force[i][1, 1] = P.coll_fx[i]; floes.collision_force[i][1, 2] = P.coll_fy[i]
    end
    unpack_status!(floes, P)
    foff = Vector{Int32}(undef, M + 1)
    check(eng, @ccall lib.sz_download_fuse(eng.ctx::Ptr{Cvoid}, foff::Ptr{Int32}, C_NULL::Ptr{Int32})::Cint)
    fidx = Vector{Int32}(undef, max(Int(foff[end]), 1))
    check(eng, @ccall lib.sz_download_fuse(eng.ctx::Ptr{Cvoid}, foff::Ptr{Int32}, fidx::Ptr{Int32})::Cint)
    for i in 1:M
        empty!(floes.status[i].fuse_idx)
        append!(floes.status[i].fuse_idx, Int.(fidx[foff[i]+1:foff[i+1]]) .+ 1)
    end
    pull_moving_boundaries!(eng, domain)
    return
end

# MovingBoundary walls moved on the device (update_boundaries!, collisions.jl:565-571): val and poly back to the host
function pull_moving_boundaries!(eng::HIPEngine, domain)
    bnds = (domain.north, domain.south, domain.east, domain.west)
    any(b -> b isa MovingBoundary, bnds) || return
    vals = Vector{Float64}(undef, 4); rects = Vector{Float64}(undef, 16)
    check(eng, @ccall lib.sz_get_boundary_vals(eng.ctx::Ptr{Cvoid}, vals::Ptr{Float64})::Cint)
    check(eng, @ccall lib.sz_get_boundary_rects(eng.ctx::Ptr{Cvoid}, rects::Ptr{Float64})::Cint)
    for (k, b) in enumerate(bnds)
        b isa MovingBoundary || continue
        b.val = vals[k]
        x0, xf, y0, yf = rects[4k-3:4k]
        b.poly = Subzero._make_bounding_box_polygon(Float64, x0, xf, y0, yf)
    end
    return
end

# timestep_coupling! (coupling.jl:1705-1738): fxOA, fyOA, trqOA, hflx_factor, status (remove: no sub-floe point in
# bounds); with two-way coupling the ice-on-ocean stress fields.  grid.floe_locations / ocean.scells are bookkeeping of
# the CPU algorithm and are not materialised (only calc_two_way_coupling!, replaced here, reads them).
function Subzero.timestep_coupling!(model::Model{Float64}, Δt, consts::Constants{Float64},
                                    coupling_settings::CouplingSettings, floe_settings::FloeSettings)
    eng = ENGINE[]
    if eng === nothing
        return invoke(Subzero.timestep_coupling!, Tuple{Any, Any, Any, Any, Any}, model, Δt, consts, coupling_settings,
                      floe_settings)
    end
    floes = model.floes
    P = upload!(eng, floes, length(floes))
    check(eng, @ccall lib.sz_timestep_coupling(eng.ctx::Ptr{Cvoid})::Cint)
    download!(eng, floes, P)
    unpack_status!(floes, P)
    coupling_settings.two_way_coupling_on && pull_ocean_stress!(eng, model.ocean)
    return
end

function pull_ocean_stress!(eng::HIPEngine, ocean)
    n = (eng.Nx + 1) * (eng.Ny + 1)
    tx, ty, sf, hf = (Vector{Float64}(undef, n) for _ in 1:4)
    check(eng, @ccall lib.sz_download_ocean_stress(eng.ctx::Ptr{Cvoid}, tx::Ptr{Float64}, ty::Ptr{Float64}, sf::Ptr{Float64},
                                                   hf::Ptr{Float64})::Cint)
    ocean.τx .= unlattice(tx, eng.Nx, eng.Ny); ocean.τy .= unlattice(ty, eng.Nx, eng.Ny)
    ocean.si_frac .= unlattice(sf, eng.Nx, eng.Ny); ocean.hflx_factor .= unlattice(hf, eng.Nx, eng.Ny)
    return
end

# timestep_floe_properties! (update_floe.jl:469-551): stress, guards, AB2 update, ring move, strain.  calc_stress! reads
# floe.interactions, which other host processes may have touched since the collisions: they go up with the columns.
function Subzero.timestep_floe_properties!(floes::StructArray{<:Floe{Float64}}, tstep, Δt, floe_settings::FloeSettings)
    eng = ENGINE[]
    if eng === nothing
        return invoke(Subzero.timestep_floe_properties!, Tuple{StructArray{<:Floe{Float64}}, Any, Any, Any}, floes, tstep, Δt,
                      floe_settings)
    end
    P = upload!(eng, floes, length(floes))
    upload_interactions!(eng, floes)
    check(eng, @ccall lib.sz_timestep_floe_properties(eng.ctx::Ptr{Cvoid}, Δt::Int32)::Cint)
    download!(eng, floes, P)
    unpack_geometry!(floes, P)
    return
end

# centroid, poly / coords and the three 2 x 2 tensors back into the per-floe objects
function unpack_geometry!(floes, P::Packed)
    for i in eachindex(floes)
        floes.centroid[i][1] = P.cx[i]; floes.centroid[i][2] = P.cy[i]
        r = P.vert_off[i]+1:P.vert_off[i+1]
        ring = [[P.vx[k], P.vy[k]] for k in r]
        floes.coords[i] = [ring]
        floes.poly[i] = Subzero.make_polygon(floes.coords[i])
        for (m, src) in ((floes.stress_accum[i], P.sa), (floes.stress_instant[i], P.si), (floes.strain[i], P.strain))
            m[1, 1], m[1, 2], m[2, 1], m[2, 2] = src[4i-3], src[4i-2], src[4i-1], src[4i]
        end
    end
    return
end

# ------------------------------------------------------------------------------------------------ resident mode
"""
    run_resident!(sim, eng; start_tstep = 0, batch = 500)

`run!(sim)` for simulations in which only the hot path touches the floes between output steps (fractures, ridging /
rafting and welding off: the defaults).  Whole batches of `timestep_sim!` run on the device with the state resident in
HBM (`sz_step`); a batch never crosses an output step of a writer, and it ends after the first step that leaves a floe
tagged `remove` / `fuse`, so `simplify_floes!` (simulation.jl:205-214) runs exactly where the reference runs it.
`sz_simplify_check` covers its other two triggers (rings over `max_vertices`, floes under the minimum area / height).
"""
function run_resident!(sim, eng::HIPEngine; start_tstep::Integer = 0, batch::Integer = 500)
    (sim.fracture_settings.fractures_on || sim.ridgeraft_settings.ridge_raft_on || sim.weld_settings.weld_on) &&
        error("run_resident!: fracture / ridging / welding need the floes on the host every step: use run!(sim)")
    Subzero.startup_sim(sim, nothing, 1)
    floes = sim.model.floes
    flags = (sim.collision_settings.collisions_on ? SZ_COLLISIONS_ON : Int32(0)) |
            (sim.coupling_settings.coupling_on ? SZ_COUPLING_ON : Int32(0))
    max_floe_id = isempty(floes) ? 0 : maximum(floes.id)
    P = upload!(eng, floes, length(floes))
    upload_interactions!(eng, floes)
    tstep, last = start_tstep, start_tstep + sim.nΔt
    dirty = false                                    # the device state is ahead of the host's
    while tstep <= last
        if output_due(sim.writers, tstep, start_tstep)
            dirty && (pull_state!(eng, floes, P); dirty = false)
            Subzero.add_ghosts!(floes, sim.model.domain)           # write_data! sees the ghosts (simulation.jl:102-105)
            Subzero.write_data!(sim, tstep, start_tstep)
            remove_ghosts!(floes)
        end
        n = min(batch, last - tstep + 1, steps_to_next_output(sim.writers, tstep, start_tstep))
        done = Ref{Int32}(0)
        check(eng, @ccall lib.sz_step(eng.ctx::Ptr{Cvoid}, n::Int32, tstep::Int32, sim.Δt::Int32,
                                      sim.coupling_settings.Δt::Int32, flags::Int32, done::Ptr{Int32})::Cint)
        tstep += done[]; dirty = true
        todo = Vector{Int64}(undef, 4)
        check(eng, @ccall lib.sz_simplify_check(eng.ctx::Ptr{Cvoid}, eng.max_vertices::Int32, eng.min_floe_area::Float64,
                                                eng.min_floe_height::Float64, todo::Ptr{Int64})::Cint)
        if any(!iszero, todo)                        # simplify_floes! has work: it runs on the host, on the full state
            pull_state!(eng, floes, P)
            max_floe_id = Subzero.simplify_floes!(sim.model, max_floe_id, sim.simp_settings, sim.collision_settings,
                                                  sim.floe_settings, sim.Δt, sim.rng)
            P = upload!(eng, floes, length(floes)); upload_interactions!(eng, floes)
            dirty = false
        end
    end
    dirty && pull_state!(eng, floes, P)
    sim.coupling_settings.two_way_coupling_on && pull_ocean_stress!(eng, sim.model.ocean)
    pull_moving_boundaries!(eng, sim.model.domain)
    Subzero.teardown_sim(sim)
    return
end

# the whole floe state back into the StructArray (columns, geometry, interactions, status + fuse lists)
function pull_state!(eng::HIPEngine, floes, P::Packed)
    M = length(floes)
    download!(eng, floes, P)
    unpack_geometry!(floes, P); unpack_status!(floes, P)
    st = stats(eng)
    off = Vector{Int32}(undef, M + 1); rows = Matrix{Float64}(undef, 7, max(Int(st.n_inter_rows), 1))
    check(eng, @ccall lib.sz_download_interactions(eng.ctx::Ptr{Cvoid}, off::Ptr{Int32}, rows::Ptr{Float64})::Cint)
    foff = Vector{Int32}(undef, M + 1)
    check(eng, @ccall lib.sz_download_fuse(eng.ctx::Ptr{Cvoid}, foff::Ptr{Int32}, C_NULL::Ptr{Int32})::Cint)
    fidx = Vector{Int32}(undef, max(Int(foff[end]), 1))
    check(eng, @ccall lib.sz_download_fuse(eng.ctx::Ptr{Cvoid}, foff::Ptr{Int32}, fidx::Ptr{Int32})::Cint)
    for i in 1:M
        r = off[i]+1:off[i+1]
        floes.interactions[i] = permutedims(rows[:, r]); floes.num_inters[i] = length(r)
        floes.collision_force[i][1, 1] = P.coll_fx[i]; floes.collision_force[i][1, 2] = P.coll_fy[i]
        empty!(floes.status[i].fuse_idx); append!(floes.status[i].fuse_idx, Int.(fidx[foff[i]+1:foff[i+1]]) .+ 1)
    end
    return
end

# ghost rows off again (simulation.jl:138-144)
function remove_ghosts!(floes)
    n = count(==(0), floes.ghost_id)
    for i in reverse(n+1:length(floes))
        StructArrays.foreachfield(f -> deleteat!(f, i), floes)
    end
    empty!.(floes.ghosts)
    return
end

# ------------------------------------------------------------------------------------------------ tiles: one process per GPU
# A tiled run (DESIGN.md §6): every rank owns the floes of one spatial tile and trades a one-deep halo of ghost-floe records
# with the neighbouring tiles each step, inside the library.  Between the ranks the library uses either RCCL (it opens the
# communicator itself from a 128-byte id the host hands round) or the host's OWN channel -- three blocking collectives on
# host memory, e.g. MPI.jl calls -- for hosts without a device-aware MPI or with several ranks on one GPU.

# include/subzero_hip.h: sz_host_transport
struct SzHostTransport
    user::Ptr{Cvoid}
    allgather::Ptr{Cvoid}
    sendrecv::Ptr{Cvoid}
    allreduce_sum_f64::Ptr{Cvoid}
end

# the host program's collectives: (allgather!(recv::Vector{UInt8}, send::Vector{UInt8}),
#                                  sendrecv!(peers::Vector{Int32}, send::Vector{Vector{UInt8}}, recv::Vector{Vector{UInt8}}),
#                                  allreduce_sum!(buf::Vector{Float64})), e.g. with MPI.jl:
#     allgather!(r, s)       = MPI.Allgather!(s, MPI.UBuffer(r, length(s)), comm)
#     sendrecv!(p, s, r)     = MPI.Waitall(vcat([MPI.Irecv!(r[k], comm; source = p[k], tag = 7) for k in eachindex(p) if !isempty(r[k])],
#                                                [MPI.Isend(s[k], comm; dest = p[k], tag = 7) for k in eachindex(p) if !isempty(s[k])]))
#     allreduce_sum!(b)      = MPI.Allreduce!(b, +, comm)
const TRANSPORT = Ref{Any}(nothing)
const TRANSPORT_NRANKS = Ref{Int}(1)

function _transport_allgather(user::Ptr{Cvoid}, send::Ptr{UInt8}, recv::Ptr{UInt8}, bytes::Int64)::Cint
    try
        TRANSPORT[][1](unsafe_wrap(Array, recv, bytes * TRANSPORT_NRANKS[]), unsafe_wrap(Array, send, bytes))
        return Cint(0)
    catch
        return Cint(1)                               # (an exception must not unwind through the C frames)
    end
end
function _transport_sendrecv(user::Ptr{Cvoid}, npeers::Int32, peer::Ptr{Int32}, send::Ptr{Ptr{UInt8}}, send_bytes::Ptr{Int64},
                             recv::Ptr{Ptr{UInt8}}, recv_bytes::Ptr{Int64})::Cint
    try
        n = Int(npeers)
        peers = unsafe_wrap(Array, peer, n); sb = unsafe_wrap(Array, send_bytes, n); rb = unsafe_wrap(Array, recv_bytes, n)
        sp = unsafe_wrap(Array, send, n); rp = unsafe_wrap(Array, recv, n)
        TRANSPORT[][2](copy(peers), [unsafe_wrap(Array, sp[k], sb[k]) for k in 1:n], [unsafe_wrap(Array, rp[k], rb[k]) for k in 1:n])
        return Cint(0)
    catch
        return Cint(1)
    end
end
function _transport_allreduce(user::Ptr{Cvoid}, buf::Ptr{Float64}, n::Int64)::Cint
    try
        TRANSPORT[][3](unsafe_wrap(Array, buf, n))
        return Cint(0)
    catch
        return Cint(1)
    end
end

"""
    tiles!(eng, nranks, rank, owned_global_index, max_ring, max_rmax, Lx, Ly; id = nothing, transport = nothing,
           periodic_x = true, periodic_y = true, drift_margin = max(2000.0, max_rmax / 2), rebox_every = 150)

Collective, after `upload!` of the floes this rank owns (those whose centroid lies in its tile; `owned_global_index[i]` =
the index floe `i` has in the whole field, `max_ring` / `max_rmax` over ALL floes: halo floes arrive unseen).  `id`: the 128
bytes of `comm_unique_id()` from rank 0, handed round by the host (RCCL between the ranks); `transport` = the three
collectives described above (the host's channel).  Then `tile_run!` replaces `sz_step`.
"""
function tiles!(eng::HIPEngine, nranks::Integer, rank::Integer, owned_global_index::Vector{Int64}, max_ring::Real, max_rmax::Real,
                Lx::Real, Ly::Real; id = nothing, transport = nothing, periodic_x::Bool = true, periodic_y::Bool = true,
                drift_margin::Real = max(2000.0, max_rmax / 2), rebox_every::Integer = 150, tile_center = nothing)
    if transport !== nothing
        TRANSPORT[] = transport; TRANSPORT_NRANKS[] = nranks
        t = Ref(SzHostTransport(C_NULL,
            @cfunction(_transport_allgather, Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Int64)),
            @cfunction(_transport_sendrecv, Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Ptr{UInt8}}, Ptr{Int64}, Ptr{Ptr{UInt8}}, Ptr{Int64})),
            @cfunction(_transport_allreduce, Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64))))
        check(eng, @ccall lib.sz_comm_init_host(eng.ctx::Ptr{Cvoid}, nranks::Int32, rank::Int32, t::Ptr{SzHostTransport})::Cint)
    else
        idp = id === nothing ? C_NULL : pointer(id)
        GC.@preserve id check(eng, @ccall lib.sz_comm_init(eng.ctx::Ptr{Cvoid}, nranks::Int32, rank::Int32, idp::Ptr{Cvoid})::Cint)
    end
    check(eng, @ccall lib.sz_tile_enable(eng.ctx::Ptr{Cvoid}, owned_global_index::Ptr{Int64}, Float64(max_ring)::Float64,
                                         Float64(max_rmax)::Float64)::Cint)
    check(eng, @ccall lib.sz_tile_setup(eng.ctx::Ptr{Cvoid}, Float64(Lx)::Float64, Float64(Ly)::Float64, Int32(periodic_x)::Int32,
                                        Int32(periodic_y)::Int32, Float64(drift_margin)::Float64, Int32(rebox_every)::Int32)::Cint)
    if tile_center !== nothing           # (x, y) of this rank's tile centre: see sz_tile_set_center
        check(eng, @ccall lib.sz_tile_set_center(eng.ctx::Ptr{Cvoid}, Float64(tile_center[1])::Float64, Float64(tile_center[2])::Float64)::Cint)
    end
    return
end

function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    @ccall(lib.sz_comm_unique_id(id::Ptr{Cvoid})::Cint) == 0 || error("sz_comm_unique_id: RCCL could not be bound")
    return id
end

# nsteps x timestep_sim! of the tiled run, collectively (same arguments on every rank).  Returns the steps run: like sz_step the batch
# ends after the first step that leaves a floe tagged remove / fuse on ANY rank (the same number on every rank), so that the host's
# simplify_floes! runs at the step the reference runs it (simulation.jl:205-214)
function tile_run!(eng::HIPEngine, nsteps::Integer, tstep::Integer, Δt::Integer, coupling_Δt::Integer, flags::Integer)
    done = Ref{Int32}(0)
    check(eng, @ccall lib.sz_tile_run(eng.ctx::Ptr{Cvoid}, nsteps::Int32, tstep::Int32, Δt::Int32, coupling_Δt::Int32, flags::Int32,
                                      done::Ptr{Int32})::Cint)
    return Int(done[])
end

# floes that left their tile go to the rank that owns the tile their centroid lies in now (px x py tiles over the domain), with their
# complete state, over the library's channel; collective, between two tile_run! calls.  Returns (floes given away, floes owned now).
function tile_migrate!(eng::HIPEngine, px::Integer, py::Integer)
    sent = Ref{Int64}(0); owned = Ref{Int64}(0)
    check(eng, @ccall lib.sz_tile_migrate(eng.ctx::Ptr{Cvoid}, px::Int32, py::Int32, C_NULL::Ptr{Int32}, sent::Ptr{Int64}, owned::Ptr{Int64})::Cint)
    return Int(sent[]), Int(owned[])
end
# global indices (0-based positions in the undivided floe list) of the floes this rank owns, after tile_migrate! too
function tile_owned(eng::HIPEngine, n_owned::Integer)
    g = Vector{Int64}(undef, max(n_owned, 1))
    check(eng, @ccall lib.sz_tile_owned_gidx(eng.ctx::Ptr{Cvoid}, g::Ptr{Int64}, length(g)::Int64)::Cint)
    return g[1:n_owned]
end

writer_periods(w) = Int[x.Δtout for ws in (w.floewriters, w.gridwriters, w.checkpointwriters) for x in ws]
output_due(w, tstep, start) = tstep == start || any(p -> mod(tstep, p) == 0, writer_periods(w))
function steps_to_next_output(w, tstep, start)
    ps = writer_periods(w)
    isempty(ps) && return typemax(Int32) ÷ 2
    return minimum(p - mod(tstep, p) for p in ps)
end


# ------------------------------------------------------------------------------------------------ Float32 hosts
# Floe{FT} is generic (floe.jl:24) although only Float64 is tested and supported (documentation.md:25).  The library's `_f32` entry points
# take the columns of a Floe{Float32} field as they lie, widen them on the way in, compute as for a Float64 host and round on the way out.
# The column traffic of such a host: `upload32!` / `download32!` below (the process-mode boundary); the overloads of the three hot calls
# above stay Float64 methods -- for a Float32 simulation they are the same bodies with these two in place of `upload!` / `download!`,
# `sz_set_fields_f32` for the lattices and `sz_download_interactions_f32` for floe.interactions.
struct Packed32
    cols::SzFloeColumnsF32
    keep::Vector{Any}                     # every flattened vector the pointers point into
    status::Vector{Int32}
end

function pack32(floes::StructArray{<:Floe{Float32}}, n_parents::Integer)
    M = length(floes)
    cx = Float32[c[1] for c in floes.centroid]; cy = Float32[c[2] for c in floes.centroid]
    coll_fx = Float32[f[1, 1] for f in floes.collision_force]; coll_fy = Float32[f[1, 2] for f in floes.collision_force]
    sa = Vector{Float32}(undef, 4M); si = similar(sa); st = similar(sa)
    for i in 1:M
        sa[4i-3:4i] .= tensor4(floes.stress_accum[i]); si[4i-3:4i] .= tensor4(floes.stress_instant[i])
        st[4i-3:4i] .= tensor4(floes.strain[i])
    end
    status = Int32[Int32(s.tag) for s in floes.status]
    vert_off = Vector{Int32}(undef, M + 1); vert_off[1] = 0
    vx = Float32[]; vy = Float32[]
    for i in 1:M
        for pt in GI.getpoint(GI.getexterior(floes.poly[i]))
            push!(vx, GI.x(pt)); push!(vy, GI.y(pt))
        end
        vert_off[i+1] = length(vx)
    end
    sub_off = Vector{Int32}(undef, n_parents + 1); sub_off[1] = 0
    sx = Float32[]; sy = Float32[]
    for i in 1:n_parents
        append!(sx, floes.x_subfloe_points[i]); append!(sy, floes.y_subfloe_points[i])
        sub_off[i+1] = length(sx)
    end
    id = Vector{Int64}(floes.id); ghost_id = Vector{Int64}(floes.ghost_id)
    c = SzFloeColumnsF32()
    c.cx = pointer(cx); c.cy = pointer(cy)
    c.rmax = pointer(floes.rmax); c.area = pointer(floes.area); c.height = pointer(floes.height)
    c.mass = pointer(floes.mass); c.moment = pointer(floes.moment); c.alpha = pointer(floes.α)
    c.u = pointer(floes.u); c.v = pointer(floes.v); c.xi = pointer(floes.ξ)
    c.p_dxdt = pointer(floes.p_dxdt); c.p_dydt = pointer(floes.p_dydt); c.p_dalphadt = pointer(floes.p_dαdt)
    c.p_dudt = pointer(floes.p_dudt); c.p_dvdt = pointer(floes.p_dvdt); c.p_dxidt = pointer(floes.p_dξdt)
    c.fxOA = pointer(floes.fxOA); c.fyOA = pointer(floes.fyOA); c.trqOA = pointer(floes.trqOA)
    c.hflx_factor = pointer(floes.hflx_factor); c.overarea = pointer(floes.overarea)
    c.coll_fx = pointer(coll_fx); c.coll_fy = pointer(coll_fy); c.coll_trq = pointer(floes.collision_trq)
    c.stress_accum = pointer(sa); c.stress_instant = pointer(si); c.strain = pointer(st)
    c.id = pointer(id); c.ghost_id = pointer(ghost_id); c.status = pointer(status)
    c.vert_off = pointer(vert_off); c.vx = pointer(vx); c.vy = pointer(vy)
    c.sub_off = pointer(sub_off); c.sx = pointer(sx); c.sy = pointer(sy)
    return Packed32(c, Any[cx, cy, coll_fx, coll_fy, sa, si, st, vert_off, vx, vy, sub_off, sx, sy, id, ghost_id], status)
end

function upload32!(eng::HIPEngine, floes::StructArray{<:Floe{Float32}}, n_parents)
    P = pack32(floes, n_parents)
    GC.@preserve P floes begin
        check(eng, @ccall lib.sz_upload_floes_f32(eng.ctx::Ptr{Cvoid}, length(floes)::Int64, n_parents::Int64,
                                                  P.cols::Ref{SzFloeColumnsF32})::Cint)
    end
    return P
end

function download32!(eng::HIPEngine, floes::StructArray{<:Floe{Float32}}, P::Packed32)
    GC.@preserve P floes begin
        check(eng, @ccall lib.sz_download_floes_f32(eng.ctx::Ptr{Cvoid}, P.cols::Ref{SzFloeColumnsF32})::Cint)
    end
    return
end

end # module
