# AlphaGPUAMD.jl — `ccall` binding of libagz for the reference's Julia host code.
#
# Drop-in for `module mcts_gpu` (mcts_gpu.jl) on AMD MI355X: same entry points, same argument meaning.
# NOT EXECUTED in this repository's CI (there is no Julia toolchain in the build image); it is the binding a
# maintainer of fabricerosay/AlphaGPU would add.  See INTEGRATION.md.
module mcts_gpu

export mcts, mcts_chain!, mcts_sharded!, duelnetwork, mcts_single, init, re_init, release_engines!

using ..Game            # Position, canPlay, play, isOver, VectorizedState, FeatureSize, maxActions, maxLengthGame, PoolSample

const libagz = get(ENV, "LIBAGZ", "libagz.so")

struct AgzConfig
    game::Int32; n::Int32; nvict::Int32; max_games::Int32; max_visits::Int32; device::Int32
    seed::UInt64; game_id_base::UInt32; nn_mode::Int32; sample_capacity_games::Int32
    reserved::NTuple{3,Int32}
end
struct AgzStats
    nsamples::Int64; total_plies::Int64; wins::Int64; draws::Int64; losses::Int64; rollouts::Int64
    plies::Int32; faults::Int32; search_seconds::Float64; total_seconds::Float64
end

mutable struct Engine
    h::Ptr{Cvoid}
    L::Int
end
check(e::Engine, rc) = rc == 0 ? nothing : error(unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), e.h)))
# an engine holds gigabytes of HBM the Julia GC cannot see: release it explicitly (the finalizer is only a safety net)
function destroy!(e::Engine)
    e.h == C_NULL || ccall((:agz_destroy, libagz), Cvoid, (Ptr{Cvoid},), e.h)
    e.h = C_NULL
end
# the reference draws unseeded CUDA.rand / StatsBase randomness on every call (mcts_gpu.jl:397,520): a fresh Philox key per call
fresh_seed() = rand(UInt64) >> 1 | UInt64(1)
set_seed!(e::Engine, seed) = check(e, ccall((:agz_set_seed, libagz), Cint, (Ptr{Cvoid}, UInt64), e.h, seed))

# ---- which game?  The reference compiles ONE game into the session: main*.jl includes a plugin module (`GoBang`, `FourIARow`, `Hex`, `RevSix`)
# and, for the boards of variable size, defines `const N` / `const Nvict` (mainGobang.jl:24-26, mainHex.jl:23) before it includes mcts_gpu.jl.
# The shim reads the same definitions, so that the reference's call sites — selfplay.jl:34 `mcts_gpu.mcts(convert_back(net), rollout,
# samplesNumber, buffer, cpuct=cpuct, noise=noise)` and :56 `mcts_gpu.duelnetwork(convert_back(trainingnet), convert_back(net), 32, 1024, -1)` —
# run UNEDITED: no game / N / Nvict argument anywhere.  agz_game_kind: 0 Gobang, 1 Connect4, 2 Hex, 3 Reversi 8x8, 4 Reversi 6x6.
function detect_game()
    M = parentmodule(@__MODULE__)                     # `Main` when the file is included the way mcts_gpu.jl is (main*.jl:82-87)
    n  = isdefined(M, :N) ? Int(getfield(M, :N)) : 0
    nv = isdefined(M, :Nvict) ? Int(getfield(M, :Nvict)) : 0
    isdefined(M, :GoBang)    && return (0, n, nv)     # Gobang.jl:1
    isdefined(M, :FourIARow) && return (1, 0, 0)      # 4IARow.jl:1
    isdefined(M, :Hex)       && return (2, n, 0)      # Hex.jl:1
    isdefined(M, :RevSix)    && return (maxActions == 65 ? 3 : 4, 0, 0)   # Reversi8x8.jl:1 / Reversi6x6.jl:1 (both `module RevSix`: 65 vs 37 actions)
    error("AlphaGPUAMD: no game plugin module (GoBang, FourIARow, Hex, RevSix) is loaded in $(M)")
end
const AGZ_GAME, AGZ_N, AGZ_NVICT = detect_game()

function init(positions::Vector{Position}, visits; game::Integer=AGZ_GAME, N::Integer=AGZ_N, Nvict::Integer=AGZ_NVICT, device=0, seed=fresh_seed(), nn_mode=0, sample_capacity=0)
    cfg = Ref(AgzConfig(game, N, Nvict, length(positions), visits, device, seed, 0, nn_mode, sample_capacity, (0, 0, 0)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:agz_create, libagz), Cint, (Ref{AgzConfig}, Ref{Ptr{Cvoid}}), cfg, h)
    rc == 0 || error(unsafe_string(ccall((:agz_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)))
    e = Engine(h[], 0)
    finalizer(destroy!, e)
    re_init(positions, e)
    e
end
init(L::Int, visits; kw...) = init([Position() for _ in 1:L], visits; kw...)

# ONE engine per (slots, visits, sample capacity, device) stays alive from generation to generation: the reference allocates its tree arrays
# on every mcts call and frees them with GC.gc(true); CUDA.reclaim() (mcts_gpu.jl:488-489) — here an engine owns gigabytes of HBM, a
# create / destroy pair per call costs ~0.3 s, and trainingPipeline calls mcts and duelnetwork once per generation with the same sizes.
# release_engines!() frees them (e.g. before the training step, if it needs the memory).
const ENGINES = Dict{NTuple{4,Int},Engine}()
function engine_for(slots::Integer, visits::Integer, capacity::Integer; device=0)
    key = (Int(slots), Int(visits), Int(capacity), Int(device))
    e = get(ENGINES, key, nothing)
    if e === nothing || e.h == C_NULL
        e = init(Int(slots), visits; device=device, sample_capacity=capacity)
        ENGINES[key] = e
    end
    e
end
function release_engines!()
    foreach(destroy!, values(ENGINES)); empty!(ENGINES)
end

# re_init(cu(positions), vnodes, L, ...) — the Vector{Position} memory image is passed as is (format 0 = Julia)
function re_init(positions::Vector{Position}, e::Engine)
    check(e, ccall((:agz_set_roots, libagz), Cint, (Ptr{Cvoid}, Ptr{Position}, Cint, Ptr{UInt32}, Cint),
                   e.h, positions, 0, C_NULL, length(positions)))
    e.L = length(positions)
end

# actor = convert_back(net)::snetwork2 (DenseNet.jl:331-333); weights are CPU Arrays in Flux (out,in) layout
function set_network!(e::Engine, actor; slot=0)
    res = isempty(actor.res) ? Float32[] : reduce(vcat, vec.(Array.(actor.res)))
    H = size(actor.base, 1)
    check(e, ccall((:agz_set_network_slot, libagz), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                   e.h, slot, H, length(actor.res), Array(actor.base), res, Array(actor.policy), Array(actor.policy_bias),
                   Array(actor.value), Array(actor.value_bias)))
end

function mcts_single(actor, visits, e::Engine; training=true, cpuct=2f0, step=0)
    actor === nothing || set_network!(e, actor)
    check(e, ccall((:agz_search, libagz), Cint, (Ptr{Cvoid}, Cint, Cfloat, Cint, UInt32), e.h, visits, cpuct, training, step))
    policy = Matrix{Float32}(undef, maxActions, e.L); batch = Matrix{Float32}(undef, 2VectorizedState, e.L)
    check(e, ccall((:agz_get_policy, libagz), Cint, (Ptr{Cvoid}, Ptr{Float32}), e.h, policy))
    check(e, ccall((:agz_get_batch, libagz), Cint, (Ptr{Cvoid}, Ptr{Float32}), e.h, batch))
    policy, batch                                  # == Array(vnodesStats.policy_final), Array(vnodesStats.batch)
end

# mcts(actor, visits, ngames, buffer; θ, cpuct, noise) — mcts_gpu.jl:477: same positional arguments, same keywords, same defaults, same
# return value.  (θ and noise are accepted and unused, as in the reference: :477 never reads θ, and expand ignores its noise argument, :250.)
# Extra keywords, all optional: seed (a fresh one per call, like the reference's unseeded draws), slots < ngames — the engine then keeps
# `slots` games in flight and a slot whose game has ended takes the next game that has not started (every search on a full batch; each
# game's samples are those of the lock-step run, agz.h agz_selfplay).
function mcts(actor, visits, ngames, buffer::PoolSample; θ=1, cpuct=2.0, noise=Float32(1 / maxActions), seed=fresh_seed(), slots=ngames, device=0)
    e = engine_for(min(slots, ngames), visits, ngames; device=device)
    set_seed!(e, seed)
    set_network!(e, actor)
    st = Ref{AgzStats}()
    rc = ccall((:agz_selfplay, libagz), Cint, (Ptr{Cvoid}, Cint, Cint, Cfloat, Cint, Ref{AgzStats}), e.h, ngames, visits, cpuct, 25, st)
    if rc == -5                                    # "faute" (mcts_gpu.jl:526-529)
        println("faute")
        return (data=[], valid=false)
    end
    check(e, rc)
    push_samples!(e, st[], buffer)
    return (data=[], valid=true)
end

# the samples of the last self-play call -> the PoolSample, in the order the reference pushes them (ply-major, slot order)
function push_samples!(e::Engine, st::AgzStats, buffer::PoolSample)
    n = st.nsamples
    state = Matrix{Int8}(undef, 2VectorizedState, n); policy = Matrix{Float32}(undef, maxActions, n)
    player = Vector{Int8}(undef, n); value = Vector{Float32}(undef, n); fstate = Matrix{Int8}(undef, FeatureSize, n)
    check(e, ccall((:agz_get_samples, libagz), Cint,
                   (Ptr{Cvoid}, Ptr{Int8}, Ptr{Float32}, Ptr{Int8}, Ptr{Float32}, Ptr{Int8}, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
                   e.h, state, policy, player, value, fstate, C_NULL, C_NULL, C_NULL))
    for i in 1:n
        idx = Main.push_buffer(buffer, state, policy, player[i], i)
        buffer.pool[idx].value = value[i]; buffer.pool[idx].fstate .= @view fstate[:, i]
    end
    println("victoires,nul,défaites", [st.wins, st.draws, st.losses])
end

# A host loop that plays generation after generation (selfplay.jl:34 inside trainingPipeline) keeps ONE engine (init(...; sample_capacity =
# ngames + next_ngames)) and calls this instead of mcts: the call returns its own ngames games and starts up to next_ngames games of the
# next call in the slots that come free meanwhile, so that no generation ends on a batch that runs out (agz.h agz_selfplay_chain; the
# last call of the loop passes next_ngames = 0).  The network may change from call to call (set_network! inside).
function mcts_chain!(e::Engine, actor, visits, ngames, next_ngames, buffer::PoolSample; cpuct=2.0)
    set_network!(e, actor)
    st = Ref{AgzStats}()
    rc = ccall((:agz_selfplay_chain, libagz), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cfloat, Cint, Ref{AgzStats}), e.h, ngames, next_ngames, visits, cpuct, 25, st)
    rc == -5 && return (data=[], valid=false)      # "faute"
    check(e, rc)
    push_samples!(e, st[], buffer)
    return (data=[], valid=true)
end

# The games a chained call starts early for the NEXT call play their first plies with the network of the running call (the reference
# plays every game of a generation with one actor, selfplay.jl:34-56).  A host loop that cares numbers its networks: every sample carries
# the tag of the network that searched it (byte 18 of a packed record, 1-based; agz.h agz_set_network_tag).  next_ngames = 0 in every
# call gives the reference's semantics exactly (each call's games are played by the call's actor only) at the price of a batch that runs
# out at the end of each call.
set_network_tag!(e::Engine, tag::Integer) = check(e, ccall((:agz_set_network_tag, libagz), Cint, (Ptr{Cvoid}, UInt32), e.h, tag))

# ---- game-id shards over several GPUs (BASELINE config 5): one Julia process per GPU, each with its own engine
# (init(...; device = local rank) and game ids from rank * ngames on: agz_config.game_id_base), nothing exchanged while games run, and ONE
# RCCL all-gather of the packed sample records at the end of the call (agz.h agz_comm_*; SURVEY 8e).  Every rank then pushes the samples
# of ALL ranks into its PoolSample — the same samples, in the same order, as one GPU playing world * ngames games.
mutable struct Comm
    c::Ptr{Cvoid}; rank::Int; world::Int
end
# id: the 128 bytes of comm_unique_id() of rank 0, shipped to the other ranks by the launcher (a file, MPI.jl, Distributed.jl)
function comm_unique_id()
    id = zeros(UInt8, 128)
    ccall((:agz_comm_unique_id, libagz), Cint, (Ptr{UInt8},), id) == 0 || error(unsafe_string(ccall((:agz_comm_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)))
    id
end
function comm_create(e::Engine, rank, world, id::Vector{UInt8}, capacity_records)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:agz_comm_create, libagz), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Int64, Ref{Ptr{Cvoid}}), e.h, rank, world, id, capacity_records, c)
    rc == 0 || error(unsafe_string(ccall((:agz_comm_last_error, libagz), Cstring, (Ptr{Cvoid},), C_NULL)))
    Comm(c[], rank, world)
end
comm_destroy!(c::Comm) = (c.c == C_NULL || ccall((:agz_comm_destroy, libagz), Cvoid, (Ptr{Cvoid},), c.c); c.c = C_NULL)

struct AgzGameInfo
    A::Int32; VS::Int32; FS::Int32; ML::Int32; max_plies::Int32; pos_image_bytes::Int32; rec_bytes::Int32; reserved::Int32
end

# mcts(actor, visits, ngames, buffer) of a rank of a sharded run: plays this rank's ngames games (a call of a chain: next_ngames as in
# mcts_chain!), gathers every rank's records and pushes them all — in PoolSample order: ply-major, then game id — into `buffer`.
function mcts_sharded!(e::Engine, comm::Comm, actor, visits, ngames, next_ngames, buffer::PoolSample; cpuct=2.0)
    set_network!(e, actor)
    st = Ref{AgzStats}()
    rc = ccall((:agz_selfplay_chain, libagz), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cfloat, Cint, Ref{AgzStats}), e.h, ngames, next_ngames, visits, cpuct, 25, st)
    # EVERY rank enters the collective, whatever its own call returned: a rank that left early (an illegal sampled move, any engine error)
    # would leave the others inside ncclAllGather for ever.  Its verdict travels with the records — agz_allgather_samples_status gathers one
    # status word per rank in front of the counts (a rank with an error sends no records) — and all ranks fail, or go on, together.
    counts = zeros(Int64, comm.world); status = zeros(Int32, comm.world)
    grc = ccall((:agz_allgather_samples_status, libagz), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int32}), e.h, comm.c, rc, counts, status)
    grc == 0 || error(unsafe_string(ccall((:agz_comm_last_error, libagz), Cstring, (Ptr{Cvoid},), comm.c)))
    if any(status .== -5)                          # "faute" on some rank: the generation is void on all of them (mcts_gpu.jl:526-529)
        println("faute")
        return (data=[], valid=false)
    end
    bad = findfirst(!=(0), status)
    bad === nothing || (bad - 1 == comm.rank ? check(e, rc) : error("rank $(bad - 1) failed with status $(status[bad])"))
    info = Ref{AgzGameInfo}()
    check(e, ccall((:agz_get_info, libagz), Cint, (Ptr{Cvoid}, Ref{AgzGameInfo}), e.h, info))
    n = sum(counts); rb = Int(info[].rec_bytes)
    records = Vector{UInt8}(undef, n * rb)
    off = 0
    for r in 0:comm.world-1
        ccall((:agz_comm_fetch_records, libagz), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt8}, Int64, Int64), comm.c, r, pointer(records, off + 1), 0, counts[r+1]) == 0 ||
            error(unsafe_string(ccall((:agz_comm_last_error, libagz), Cstring, (Ptr{Cvoid},), comm.c)))
        off += counts[r+1] * rb
    end
    state = Matrix{Int8}(undef, 2VectorizedState, n); policy = Matrix{Float32}(undef, maxActions, n)
    player = Vector{Int8}(undef, n); value = Vector{Float32}(undef, n); fstate = Matrix{Int8}(undef, FeatureSize, n)
    gid = Vector{UInt32}(undef, n); ply = Vector{Int32}(undef, n)
    ccall((:agz_unpack_records, libagz), Cint,
          (Ref{AgzGameInfo}, Ptr{UInt8}, Int64, Ptr{Int8}, Ptr{Float32}, Ptr{Int8}, Ptr{Float32}, Ptr{Int8}, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
          info, records, n, state, policy, player, value, fstate, gid, ply, C_NULL) == 0 || error("agz_unpack_records failed")
    for i in sortperm(collect(zip(ply, gid)))      # the order one engine playing all the games would have pushed them in
        idx = Main.push_buffer(buffer, state, policy, player[i], i)
        buffer.pool[idx].value = value[i]; buffer.pool[idx].fstate .= @view fstate[:, i]
    end
    return (data=[], valid=true)
end

# duelnetwork(actor1, actor2, visits, ngames, conv=2) — mcts_gpu.jl:653-668: the same five positional arguments (conv is accepted and unused
# here: the network family is fixed to snetwork2, SURVEY 8b), returns (v, n, d) from actor1's point of view.  Both halves reuse ONE engine.
function duelnetwork(actor1, actor2, visits, ngames, conv=2; seed=fresh_seed(), device=0)
    half = div(ngames, 2)
    e = engine_for(half, visits, half; device=device)
    function half_duel(a, b, sd)
        set_seed!(e, sd)
        set_network!(e, a; slot=0); set_network!(e, b; slot=1)
        wdl = zeros(Int64, 3)
        check(e, ccall((:agz_duel, libagz), Cint, (Ptr{Cvoid}, Cint, Cint, Cfloat, Cint, Cint, Ptr{Int64}), e.h, half, visits, 2f0, 15, 0, wdl))
        return wdl
    end
    v1, n1, d1 = half_duel(actor1, actor2, seed)
    d2, n2, v2 = half_duel(actor2, actor1, seed + 1)
    v1 + v2, n1 + n2, d1 + d2
end

end # module
