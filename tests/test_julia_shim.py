"""The Julia binding (decentralopf.jl_amd/julia/DecentralOPFHip.jl) cannot be executed here (no Julia in the image), so
what can be checked mechanically is: its C struct mirrors against include/dopf.h field by field, every ccall's symbol,
return type and arity against the header's prototypes, and the structural points that make it a drop-in for
src/imports.jl + src/structures/admm.jl of the reference."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "decentralopf.jl_amd", "julia", "DecentralOPFHip.jl")).read()
HDR = open(os.path.join(ROOT, "include", "dopf.h")).read()
CODE = "\n".join(ln.split("#", 1)[0] if not ln.lstrip().startswith('"""') else ln for ln in JL.splitlines())    # comments off

C2JL = {"int32_t": "Cint", "double": "Cdouble", "const double *": "Ptr{Cdouble}", "const int32_t *": "Ptr{Cint}",
        "void *": "Ptr{Cvoid}"}


def c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HDR, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const double \*|const int32_t \*|void \*|int32_t|double)\s*(.*)", decl)
        ctype, names = m.group(1).strip(), m.group(2)
        if ctype.endswith("*"):
            ctype = ctype[:-1].strip() + " *"
        for n in names.split(","):
            out.append((n.strip().lstrip("*"), C2JL[ctype]))
    return out


def jl_struct_fields(name):
    body = re.search(r"^struct %s[ \t]*\n(.*?)^end" % name, CODE, re.S | re.M).group(1)
    out = []
    for part in re.split(r"[;\n]", body):
        part = part.strip()
        if part:
            n, t = part.split("::")
            out.append((n.strip(), t.strip()))
    return out


def test_c_struct_mirrors_match_the_header():
    assert jl_struct_fields("CProblem") == c_struct_fields("dopf_problem")
    assert jl_struct_fields("CParams") == c_struct_fields("dopf_params")


def header_prototypes():
    text = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[\w]+\s*\**)\s*(dopf_\w+)\s*\(([^;{]*?)\)\s*;", text, re.M):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        n = 0 if args in ("", "void") else len(args.split(","))
        protos[name] = (ret.replace(" *", "*"), n)
    return protos


RET = {"int": "Cint", "void": "Cvoid", "const char*": "Cstring", "dopf_ctx*": "Ptr{Cvoid}", "int32_t": "Cint", "int64_t": "Clonglong"}


def test_every_ccall_matches_a_header_prototype():
    protos = header_prototypes()
    calls = re.findall(r"ccall\(\(:(\w+), DOPF_LIB\),\s*([\w{}]+),\s*\(([^)]*)\)", CODE)
    assert len(calls) >= 12
    for sym, ret, args in calls:
        assert sym in protos, f"{sym} is not declared in include/dopf.h"
        cret, nargs = protos[sym]
        assert RET[cret] == ret, (sym, cret, ret)
        got = len([a for a in args.split(",") if a.strip()])
        assert got == nargs, (sym, got, nargs)
    used = {c[0] for c in calls}
    for need in ("dopf_create", "dopf_destroy", "dopf_iterate", "dopf_last_error", "dopf_get_duals", "dopf_get_duals_used",
                 "dopf_get_primal", "dopf_get_consensus", "dopf_get_residuals", "dopf_multi_create", "dopf_multi_iterate",
                 "dopf_multi_destroy"):
        assert need in used, need


def header_arg_types():
    """name -> list of C argument types (parameter names and const stripped, '*' attached)"""
    text = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*(?:const\s+)?[\w]+\s*\**\s*(dopf_\w+)\s*\(([^;{]*?)\)\s*;", text, re.M):
        name, args = m.group(1), " ".join(m.group(2).split())
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip().replace("const ", "")
                mm = re.match(r"([\w]+)\s*(\**)\s*\w*$", a)
                assert mm, (name, a)
                types.append(mm.group(1) + mm.group(2))
        out[name] = types
    return out


# what a ccall may declare for a C argument type (Ref{T} = a scalar passed by reference, Ptr{T} = an array)
ARG = {
    "int32_t": {"Cint"}, "double": {"Cdouble"}, "int64_t": {"Clonglong"},
    "double*": {"Ptr{Cdouble}", "Ref{Cdouble}"}, "int32_t*": {"Ptr{Cint}", "Ref{Cint}"},
    "void*": {"Ptr{Cvoid}"}, "uint64_t*": {"Ptr{UInt64}"},
    "dopf_ctx*": {"Ptr{Cvoid}"}, "dopf_multi*": {"Ptr{Cvoid}"},
    "dopf_ctx**": {"Ref{Ptr{Cvoid}}"}, "dopf_multi**": {"Ref{Ptr{Cvoid}}"},
    "dopf_problem*": {"Ref{CProblem}"}, "dopf_params*": {"Ref{CParams}"}, "dopf_central_result*": {"Ref{CCentralResult}"},
}


def split_top(s):
    """split a Julia type tuple at top-level commas (Ref{Ptr{Cvoid}} holds braces)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_every_ccall_argument_type_matches_the_header():
    """Not only the arity: every argument of every ccall is declared with a Julia type that has the C parameter's size and
    meaning (Cint for int32_t, Ptr/Ref{Cdouble} for double *, Ref{CProblem} for const dopf_problem *, ...)."""
    want = header_arg_types()
    calls = re.findall(r"ccall\(\(:(\w+), DOPF_LIB\),\s*[\w{}]+,\s*\(((?:[^()]|\([^()]*\))*?)\)\s*,", CODE, re.S)
    assert len(calls) >= 12
    for sym, args in calls:
        got = split_top(" ".join(args.split()))
        assert len(got) == len(want[sym]), (sym, got, want[sym])
        for i, (g, w) in enumerate(zip(got, want[sym])):
            assert g in ARG[w], f"{sym}: argument {i + 1} is {w} in include/dopf.h, the ccall says {g}"
    assert "dopf_get_node_results" in {c[0] for c in calls} and "dopf_central_solve" in {c[0] for c in calls}


def test_export_results_is_there_and_needs_the_history():
    assert re.search(r"function export_results\(admm::ADMM, filename::String", CODE)
    assert "iteration,dual,timestep,line,value" in CODE and "iteration,generator,timestep,generation" in CODE
    assert "iteration,storage,timestep,charge,discharge" in CODE
    assert re.search(r"admm\.record \|\| error", CODE)


def test_central_result_mirror_matches_the_header():
    body = re.search(r"typedef struct dopf_central_result \{(.*?)\} dopf_central_result;", HDR, re.S).group(1)
    want = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if decl:
            ctype, names = decl.split(" ", 1)
            want += [(n.strip(), C2JL[ctype]) for n in names.split(",")]
    assert jl_struct_fields("CCentralResult") == want


def test_it_is_a_drop_in_for_imports_jl():
    # evaluated in Main like the reference's imports.jl: no module around the type definitions, so that
    # cases/three_node.jl constructs the very Node/Generator/Storage/Line types ADMM(...) accepts
    assert not re.search(r"^\s*module\s", CODE, re.M)
    assert re.search(r"^using LinearAlgebra", CODE, re.M)                     # calculate_ptdf needs Diagonal
    assert "Gurobi" not in CODE and "JuMP" not in CODE and 'include("imports.jl")' not in CODE
    assert re.search(r'include\(joinpath\(DOPF_SRC, "structures", "network_elements.jl"\)\)', CODE)
    assert re.search(r'include\(joinpath\(DOPF_SRC, "helpers", "ptdf.jl"\)\)', CODE)
    # none of the reference's element types is re-declared here
    for t in ("Node", "Generator", "Storage", "Line"):
        assert not re.search(r"struct %s\b" % t, CODE), t
    # the reference's names and argument lists (src/structures/admm.jl:23-27, src/optimization/run.jl:1,7,
    # src/helpers/network_elements.jl:16)
    assert re.search(r"function ADMM\(gamma::Float64, nodes::Vector\{Node\}, generators::Vector\{Generator\}, storages::Vector\{Storage\},\s*"
                     r"lines::Vector\{Line\};", CODE)
    assert re.search(r"function run!\(admm::ADMM", CODE) and re.search(r"function calculate_iteration!\(admm::ADMM", CODE)
    assert re.search(r"function get_nodal_price\(iteration::Int\)", CODE)
    # the fields a caller of the reference reads
    body = re.search(r"^mutable struct ADMM\n(.*?)^end", CODE, re.S | re.M).group(1)
    for f in ("iteration::Int", "gamma::Float64", "lambdas::Vector{Vector{Float64}}", "mues::Vector{Matrix{Float64}}",
              "rhos::Vector{Matrix{Float64}}", "convergence::Convergence", "ptdf::Matrix{Float64}", "f_max::Vector{Float64}"):
        assert f in body, f
    # node ids go over 0-based
    assert "node_to_id[g.node] - 1" in CODE and "node_to_id[s.node] - 1" in CODE
