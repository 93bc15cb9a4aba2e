# Times the reference package itself (Subzero.jl, only if it is installed on the machine bench.py runs on) on the
# floe field bench.py wrote as text -- the "reference" kind of cpu_baseline (SURVEY §8d).  It contains no reference
# source: it builds the model through the package's public constructors (as examples/uniform_flow.jl does) and
# steps it with timestep_sim!.  Usage: julia -t <threads> bench/subzero_cpu.jl field.txt nsteps
#
# field.txt: line 1 = n L dt dgrid hmean E uocn npoint_per_cell, then per floe: nv u v xi x1 y1 ... x_nv y_nv
# (closed ring, clockwise).  Fractures, ridging/rafting, welding off (their defaults); vertex smoothing off.
using Subzero, StructArrays, Random

lines = readlines(ARGS[1]); nsteps = parse(Int, ARGS[2])
h = split(lines[1])
n = parse(Int, h[1]); L = parse(Float64, h[2]); dt = parse(Int, h[3]); dgrid = parse(Float64, h[4])
hmean = parse(Float64, h[5]); E = parse(Float64, h[6]); uocn = parse(Float64, h[7]); npc = parse(Int, h[8])
grid = RegRectilinearGrid(; x0 = 0.0, xf = L, y0 = 0.0, yf = L, Δx = dgrid, Δy = dgrid)
ocean = Ocean(; grid, u = uocn, v = 0.0, temp = 0.0)
atmos = Atmos(; grid, u = 0.0, v = 0.0, temp = 0.0)
domain = Domain(; north = PeriodicBoundary(North; grid), south = PeriodicBoundary(South; grid),
                east = PeriodicBoundary(East; grid), west = PeriodicBoundary(West; grid))
fs = FloeSettings(subfloe_point_generator = SubGridPointsGenerator(grid, npc))
floes = StructArray([begin
    t = parse.(Float64, split(l)); nv = Int(t[1])
    coords = [[[t[5 + 2k], t[6 + 2k]] for k in 0:nv-1]]
    Floe(coords, hmean, 0.0; floe_settings = fs, u = t[2], v = t[3], ξ = t[4])
end for l in lines[2:end]])
model = Model(grid, ocean, atmos, domain, floes)
sim = Simulation(model = model, consts = Constants(E = E), Δt = dt, nΔt = nsteps + 3, floe_settings = fs,
                 coupling_settings = CouplingSettings(Δt = 1),
                 simp_settings = SimplificationSettings(smooth_vertices_on = false))
for t in 0:2                       # compile + warm up
    Subzero.timestep_sim!(sim, t)
end
t0 = time()
for t in 3:(2 + nsteps)
    Subzero.timestep_sim!(sim, t)
end
el = time() - t0
println("RESULT floe_steps_per_sec=", n * nsteps / el, " threads=", Threads.nthreads(), " steps=", nsteps, " julia=", VERSION)
