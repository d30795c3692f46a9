# EcneHIP.jl — Julia shim over libecne_hip (C ABI: include/ecne.h).
#
# Drop-in for the three functions that form Ecne's seam for the solver path:
#   readR1CS(filename)                     (reference src/ParseR1CS.jl:50)
#   SolveConstraintsSymbolic(...)          (reference src/R1CSConstraintSolver.jl:583)
#   solveWithTrustedFunctions(...)         (reference src/R1CSConstraintSolver.jl:502)
# Same names, argument meaning, return values and exception types. All propagation work runs in
# the HIP kernels; this file only marshals handles. Julia is not available in the build image, so
# this shim has not been executed there; it mirrors ecneproject_amd/_lib.py (ctypes) call for call.
module EcneHIP

export readR1CS, SolveConstraintsSymbolic, solveWithTrustedFunctions, EcneSystem, EcneR1CS

const LIB = get(ENV, "ECNE_HIP_LIB", joinpath(@__DIR__, "..", "ecneproject_amd", "libecne_hip.so"))

struct EcneOpts
    device::Int32; secp_solve::Int32; debug::Int32; queue_mode::Int32; stream::Ptr{Cvoid}
end
struct EcneSummary
    status::Int32; function_good::Int32
    unique_nontrivial::Int64; n_nontrivial::Int64; unique_targets::Int64; n_targets::Int64
    successful_steps::Int64; outer_iterations::Int64; pops::Int64; num_unique::Int64
    rule_hits::NTuple{16,Int64}; n_rows::Int64; n_vars::Int64; pop_nnz::Int64
    device_ms::Float64; classify_ms::Float64; queue_ms::NTuple{8,Float64}; multi_ms::NTuple{8,Float64}; phase_ms::NTuple{8,Float64}
end
struct EcneInfo
    field_size::UInt32; n_wires::UInt32; n_pub_out::UInt32; n_pub_in::UInt32; n_prv_in::UInt32
    n_constraints::UInt32; n_labels::UInt64; nnz::NTuple{3,UInt64}; n_vars::Int64
end

mutable struct EcneR1CS            # the `equations` value readR1CS returns
    h::Ptr{Cvoid}
    function EcneR1CS(path::String)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ecne_r1cs_load, LIB), Cint, (Cstring, Ref{Ptr{Cvoid}}), path, out))
        x = new(out[]); finalizer(o -> ccall((:ecne_r1cs_free, LIB), Cvoid, (Ptr{Cvoid},), o.h), x); x
    end
end
mutable struct EcneSystem          # rows after abstraction + special constraints + I/O lists
    h::Ptr{Cvoid}
    function EcneSystem(f::EcneR1CS)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ecne_system_from_r1cs, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), f.h, out))
        x = new(out[]); finalizer(o -> ccall((:ecne_system_free, LIB), Cvoid, (Ptr{Cvoid},), o.h), x); x
    end
end

# status codes -> the exception the reference raises at the corresponding site
function check(st::Integer)
    st == 0 && return
    msg = unsafe_string(ccall((:ecne_strerror, LIB), Cstring, (Cint,), st))
    st == -1 && throw(AssertionError(msg))           # @assert in readR1CS (ParseR1CS.jl:58,62,69)
    st == -2 && throw(BoundsError())                 # variable_states[-1] (:916), special indexing (:762,:785)
    st == -3 && throw(DivideError())                 # divexact by zero (:919-920, :1467)
    st == -4 && throw(UndefVarError(:dsu))           # :762 without secp_solve
    st == -5 && throw(KeyError(msg))                 # abstraction's variable map (:381-382)
    st == -7 && throw(SystemError(msg))
    error("ecne_hip: $msg (status $st)")
end

function readR1CS(filename::String)                  # -> (equations, known, outputs, nVars)
    f = EcneR1CS(filename)
    kn = Ref{Ptr{Int64}}(C_NULL); nk = Ref{Csize_t}(0); tg = Ref{Ptr{Int64}}(C_NULL); nt = Ref{Csize_t}(0)
    check(ccall((:ecne_r1cs_io, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Int64}}, Ref{Csize_t}, Ref{Ptr{Int64}}, Ref{Csize_t}),
                f.h, kn, nk, tg, nt))
    info = Ref{EcneInfo}()
    check(ccall((:ecne_r1cs_info, LIB), Cint, (Ptr{Cvoid}, Ref{EcneInfo}), f.h, info))
    return f, copy(unsafe_wrap(Array, kn[], Int(nk[]))), copy(unsafe_wrap(Array, tg[], Int(nt[]))), Int64(info[].n_vars)
end

function SolveConstraintsSymbolic(constraints, special_constraints=Any[], known_variables=Int64[],
                                  debug::Bool=false, target_variables=Int64[], num_variables::Int=-1,
                                  input_sym::String="default.sym", secp_solve::Bool=false; device::Int=0)
    sys = constraints isa EcneSystem ? constraints : EcneSystem(constraints)
    opts = Ref(EcneOpts(device, secp_solve, 0, 0, C_NULL))
    res = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ecne_solve, LIB), Cint, (Ptr{Cvoid}, Ref{EcneOpts}, Ref{Ptr{Cvoid}}), sys.h, opts, res))
    s = Ref{EcneSummary}()
    check(ccall((:ecne_result_summary, LIB), Cint, (Ptr{Cvoid}, Ref{EcneSummary}), res[], s))
    ccall((:ecne_result_free, LIB), Cvoid, (Ptr{Cvoid},), res[])
    check(s[].status)
    println("Solved for ", s[].unique_nontrivial, " variables out of ", s[].n_nontrivial, " total variables")           # :1565
    println("Solved for ", s[].unique_targets, " target variables out of ", s[].n_targets, " total target variables")  # :1586
    println("------ Bad Constraints ------"); println()
    return s[].function_good == 1
end

function solveWithTrustedFunctions(input_r1cs::String, input_r1cs_name::String;
        trusted_r1cs::Vector{String}=String[], trusted_r1cs_names::Vector{String}=String[], debug::Bool=false,
        printRes::Bool=true, abstractionOnly::Bool=false, input_sym::String="", secp_solve::Bool=false)
    @assert length(trusted_r1cs) == length(trusted_r1cs_names)
    main, _, _, _ = readR1CS(input_r1cs)
    fl = [(trusted_r1cs_names[i], EcneR1CS(trusted_r1cs[i])) for i in 1:length(trusted_r1cs)]
    ncons(f) = (i = Ref{EcneInfo}(); ccall((:ecne_r1cs_info, LIB), Cint, (Ptr{Cvoid}, Ref{EcneInfo}), f.h, i); Int(i[].n_constraints))
    fl = sort(fl, by = x -> -ncons(x[2]))                                   # :527
    sys = EcneSystem(main)
    for (name, f) in fl                                                     # :531-544
        printRes && println("called abstraction")
        check(ccall((:ecne_abstract, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cstring), sys.h, f.h, name))
    end
    abstractionOnly && return true
    result = SolveConstraintsSymbolic(sys, Any[], Int64[], debug, Int64[], -1, input_sym, secp_solve)
    if result
        if !isempty(fl)
            printRes && throw(UndefVarError(:msg))                          # the reference's :556-559 behaviour
            return true
        end
        printRes && println("R1CS function " * input_r1cs_name * " has sound constraints (No trusted functions needed!)")
        return true
    end
    printRes && println("R1CS function " * input_r1cs_name * " has potentially unsound constraints")
    return false
end

end # module
