/++
mir.optim.boxcqp over the MI355X library: settings / status PODs of the bound-constrained QP that every LM pass solves
(they are embedded in `LeastSquaresSettings`, so their layout is part of the C ABI), the workspace-length functions of
the reference's C tier (boxcqp.d lines 31-51) as prototypes, and `solveBoxQP` forwarding to the device solver
`mir_solve_box_qp_gpu_d/_s` (the reference only has D overloads for it, lines 85-379).

NOT COMPILED in the build image (no D toolchain); see least_squares.d in this directory.
+/
module mir.optim.boxcqp;

version (mir_optim_amd):

import mir.ndslice.slice: Slice, Canonical;

/// Exit status of solveBoxQP; values are part of the C ABI.
enum BoxQPStatus : int
{
    solved = 0,
    numericError = 1,
    maxIterations = 2,
}

/// Tolerances of the active-set classification and the iteration limit (0: 10 n + 100).
struct BoxQPSettings(T)
    if (is(T == float) || is(T == double))
{
    T relTolerance = T.epsilon * 16;
    T absTolerance = T.epsilon * 16;
    uint maxIterations = 0;
}

static assert(BoxQPSettings!double.sizeof == 24 && BoxQPSettings!float.sizeof == 12);

extern(C) @safe pure nothrow @nogc
{
    size_t mir_box_qp_work_length(size_t n);     /// 2 n^2 + 8 n
    size_t mir_box_qp_iwork_length(size_t n);    /// n + ceil(n / 4)
}

extern(C) @system nothrow @nogc pure
{
    int mir_solve_box_qp_gpu_d(scope const BoxQPSettings!double* settings, size_t n, const(double)* P, const(double)* q,
        const(double)* l, const(double)* u, double* x, int unconstrainedSolution, int* iterations);
    int mir_solve_box_qp_gpu_s(scope const BoxQPSettings!float* settings, size_t n, const(float)* P, const(float)* q,
        const(float)* l, const(float)* u, float* x, int unconstrainedSolution, int* iterations);
}

/++
argmin_x (x'Px / 2 + q'x) subject to l <= x <= u, P positive definite with its LOWER triangle meaningful (row-major,
row stride = P's leading dimension must equal n: pass a contiguous matrix). `unconstrainedSolution`: x already holds
the unconstrained minimiser. Work slices are accepted for signature compatibility and not used.
+/
BoxQPStatus solveBoxQP(T)(
    ref const BoxQPSettings!T settings,
    Slice!(T*, 2, Canonical) P, Slice!(const(T)*) q, Slice!(const(T)*) l, Slice!(const(T)*) u, Slice!(T*) x,
    bool unconstrainedSolution = false,
    Slice!(T*) work = Slice!(T*).init, Slice!(int*) iwork = Slice!(int*).init, bool restoreUpperP = true) @trusted pure nothrow @nogc
    if (is(T == float) || is(T == double))
{
    const n = q.length;
    assert(P.length!0 == n && P.length!1 == n && P._stride!0 == n, "solveBoxQP: P must be a contiguous n x n matrix");
    assert(l.length == n && u.length == n && x.length == n);
    static if (is(T == double))
        const st = mir_solve_box_qp_gpu_d(&settings, n, P.ptr, q.ptr, l.ptr, u.ptr, x.ptr, unconstrainedSolution, null);
    else
        const st = mir_solve_box_qp_gpu_s(&settings, n, P.ptr, q.ptr, l.ptr, u.ptr, x.ptr, unconstrainedSolution, null);
    return cast(BoxQPStatus) st;
}

/// ditto, default settings first-argument-free overload (reference lines 85-102)
BoxQPStatus solveBoxQP(T)(
    Slice!(T*, 2, Canonical) P, Slice!(const(T)*) q, Slice!(const(T)*) l, Slice!(const(T)*) u, Slice!(T*) x,
    BoxQPSettings!T settings = BoxQPSettings!T.init) @trusted pure nothrow @nogc
    if (is(T == float) || is(T == double))
{
    return solveBoxQP(settings, P, q, l, u, x, false);
}

/// x = min(max(x, l), u) elementwise
void applyBounds(T)(Slice!(T*) x, Slice!(const(T)*) l, Slice!(const(T)*) u) @safe pure nothrow @nogc
{
    import mir.math.common: fmin, fmax;
    foreach (i; 0 .. x.length)
        x[i] = x[i].fmin(u[i]).fmax(l[i]);
}
