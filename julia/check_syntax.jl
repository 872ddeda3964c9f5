# CI-style check for boxes that have a `julia` binary but not Crux.jl: parses the binding and the reference probe without loading any package.
#   julia julia/check_syntax.jl
for f in ("CruxHIP.jl", "crux_ref_bench.jl")
    src = read(joinpath(@__DIR__, f), String)
    ex = Meta.parseall(src; filename=f)
    bad = [a for a in ex.args if a isa Expr && a.head in (:error, :incomplete)]
    isempty(bad) || (foreach(println, bad); error("$f does not parse"))
    println(f, ": ok (", count(==('\n'), src), " lines)")
end
