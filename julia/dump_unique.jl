# dump_unique.jl — for anyone who HAS Julia 1.7 and an instantiated Ecne checkout: run the reference solver itself and
# dump, per variable, what it ended up knowing, so that the HIP engine's result can be compared with the real thing
# (tests/tools/compare_julia_dump.py), or dropped into tests/golden/julia/ where tests/test_julia_dumps.py picks it up and compares it
# with the oracle and with the second reading (tests/ref2.py) on the CPU. Never needed by the test-suite: the build image has no Julia, which is why the
# per-variable state is "parity unpinned" against the reference (DESIGN.md §2) until somebody runs this.
#
#   julia --project=<Ecne checkout> julia/dump_unique.jl <Ecne checkout> main.r1cs out.tsv [--secp] [trusted.r1cs Name]...
#
# The script is this build's own code; it calls the checkout's exported functions (src/R1CSConstraintSolver.jl:1651) and
# re-declares nothing of it. SolveConstraintsSymbolic keeps `variable_states` local, so the solver file is evaluated with
# one line appended to its final `return` that stores the vector in a global — the checkout itself is not modified.
ecne, main, out = ARGS[1], ARGS[2], ARGS[3]
rest = ARGS[4:end]
secp = "--secp" in rest
rest = filter(x -> x != "--secp", rest)
trusted = rest[1:2:end]; names = rest[2:2:end]
src = read(joinpath(ecne, "src", "R1CSConstraintSolver.jl"), String)
hook = "    global ECNE_DUMP_STATES = variable_states\n    return function_good\nend"
occursin("    return function_good\nend", src) || error("solver source does not end SolveConstraintsSymbolic the way this script expects")
src = replace(src, "    return function_good\nend" => hook; count = 1)
cd(joinpath(ecne, "src")) do
    include_string(Main, src, joinpath(ecne, "src", "R1CSConstraintSolver.jl"))
end
verdict = Main.R1CSConstraintSolver.solveWithTrustedFunctions(main, "dump"; trusted_r1cs = trusted, trusted_r1cs_names = names,
                                                               printRes = false, secp_solve = secp)
st = Main.R1CSConstraintSolver.ECNE_DUMP_STATES
open(out, "w") do io
    println(io, "# input\t", basename(main), "\t", Int(secp), "\t", join([basename(t) * "=" * n for (t, n) in zip(trusted, names)], ","))
    println(io, "# verdict\t", verdict)
    println(io, "# var\tunique\tis_known\tlb\tub\tabz\tvalues")
    for (i, s) in enumerate(st)
        println(io, i, "\t", Int(s.unique), "\t", Int(s.is_known), "\t", s.lb.d, "\t", s.ub.d, "\t", s.abz, "\t", join(sort!([v.d for v in s.values]), ","))
    end
end
println("wrote ", out, " (", length(st), " variables, verdict ", verdict, ")")
