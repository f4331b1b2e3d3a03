/++
mir.optim.least_squares over the MI355X library (libmir_optim_amd.so): the D half of the drop-in.

This module keeps the PUBLIC surface of libmir/mir-optim's `source/mir/optim/least_squares.d` -- the status enum, the
settings / result PODs, the callback aliases, the four API tiers (`optimize`, `optimizeLeastSquares`, the precompiled
`optimizeLeastSquaresD/S`, the `extern(C)` tier) -- and contains NO Levenberg-Marquardt arithmetic: the core
(reference lines 877-1176) and the BOXCQP solve it calls live in the HIP library, whose C ABI (include/mir_optim_amd.h,
part 1) is exactly the reference's `extern(C)` tier. So the direction of the reference's own adaptor
(`optimizeLeastSquaresImplGenericBetterC`, reference lines 803-867: C function pointers -> D delegates) is inverted
here: D delegates -> C function pointers, then `mir_optimize_least_squares_d/_s`.

Status: NOT COMPILED in the build image (no ldc2 / dmd / gdc / dub there). The struct layouts this file declares are
pinned from the C side (static_asserts in mir_optim_amd/csrc/abi.hip, tests/test_abi.py, tests/test_c_harness.py):
  LeastSquaresSettings!double 128 bytes (qpSettings at 104), !float 68; LeastSquaresResult!double 32, !float 24;
  Slice!(T*) = { size_t length; T* ptr }; LeastSquaresTask = 16-byte delegate { context, funcptr };
  LeastSquaresStatus / BoxQPStatus 32-bit enums; lapackint = int (the `*-ilp` configurations: dlang/README.md, "ILP64").
BetterC: nothing here allocates with the GC or throws except `optimize` under version(D_Exceptions).

Build: add this directory's `source` in front of mir-optim's own (or replace the two files) and link the library:
    dub.sdl:   versions "mir_optim_amd" ; libs "mir_optim_amd" ; lflags "-L$MIR_OPTIM_AMD/lib" "-rpath=$MIR_OPTIM_AMD/lib"
+/
module mir.optim.least_squares;

version (mir_optim_amd):

import mir.ndslice.slice: Slice, sliced;
import mir.optim.boxcqp: BoxQPSettings;

/// 32-bit integers in `iwork` (the library never dereferences `iwork`; the type only sizes the caller's buffer)
alias lapackint = int;

/// Exit status of a solve; the numeric values are part of the C ABI (reference lines 20-46).
enum LeastSquaresStatus : int
{
    maxIterations = -1,      /// iteration limit reached
    furtherImprovement = 0,  /// lambda exceeded maxLambda: no better point can be found
    xConverged = 1,          /// step below the tolerances
    gConverged = 2,          /// gradient below gradTolerance
    fConverged = 3,          /// residual below maxGoodResidual
    badBounds = -32,
    badGuess = -31,
    badMinStepQuality = -30,
    badGoodStepQuality = -29,
    badStepQuality = -28,
    badLambdaParams = -27,
    numericError = -26,
}

/// D-tier callbacks (delegates over ndslice views).
alias LeastSquaresFunction(T) = void delegate(Slice!(const(T)*) x, Slice!(T*) y) @safe nothrow @nogc pure;
/// ditto (J is row-major m x n)
alias LeastSquaresJacobian(T) = void delegate(Slice!(const(T)*) x, Slice!(T*, 2) J) @safe nothrow @nogc pure;
/// C-tier callbacks (function pointer + opaque context).
alias LeastSquaresFunctionBetterC(T) = extern(C) void function(scope void* context, size_t m, size_t n, const(T)* x, T* y) @system nothrow @nogc pure;
/// ditto
alias LeastSquaresJacobianBetterC(T) = extern(C) void function(scope void* context, size_t m, size_t n, const(T)* x, T* J) @system nothrow @nogc pure;

/// One finite-difference column; the thread manager must run it for every `i` in `[0, count)`.
alias LeastSquaresTask = void delegate(uint totalThreads, uint threadId, uint i) @safe nothrow @nogc pure;
/// The same as a C function over the (opaque, 16-byte, by-value) task.
alias LeastSquaresTaskBetterC = extern(C) void function(scope const LeastSquaresTask, uint totalThreads, uint threadId, uint i) @safe nothrow @nogc pure;
/// Spreads `count` tasks over threads (D tier).
alias LeastSquaresThreadManager = void delegate(uint count, scope LeastSquaresTask task) @safe nothrow @nogc pure;
/// ditto (C tier)
alias LeastSquaresThreadManagerBetterC = extern(C) void function(scope void* context, uint count, scope const LeastSquaresTask taskContext, scope LeastSquaresTaskBetterC task) @system nothrow @nogc pure;

/++
Iteration settings. Field order, types and defaults are the reference's (lines 85-123); the defaults are ALSO what
`mir_least_squares_init_d/_s` of the library write, so `LeastSquaresSettings!T.init` and an initialised C struct agree.
+/
struct LeastSquaresSettings(T)
    if (is(T == double) || is(T == float))
{
    import mir.math.common: sqrt;
    import mir.math.constant: GoldenRatio;

    uint maxIterations = 1000;                              /// accepted steps allowed
    uint maxAge;                                            /// Jacobian age limit; 0: 3 with an analytic Jacobian, else 2n
    T jacobianEpsilon = T(2) ^^ ((1 - T.mant_dig) / 2);     /// ABSOLUTE finite-difference step (2^-26 / 2^-11: integer division)
    T absTolerance = T.epsilon;                             /// |dx|_2 <= absTolerance: x converged
    T relTolerance = 0;                                     /// |x|_2 <= |dx|_2 relTolerance: x converged (sic)
    T gradTolerance = T.epsilon;                            /// |J^T y|_inf
    T maxGoodResidual = T.epsilon ^^ 2;                     /// sum of squares small enough
    T maxStep = T.max.sqrt / 16;
    T maxLambda = T.max / 16;
    T minLambda = T.min_normal * 16;
    T minStepQuality = 0.1;
    T goodStepQuality = 0.5;
    T lambdaIncrease = 2;
    T lambdaDecrease = 1 / (GoldenRatio * 2);
    BoxQPSettings!T qpSettings;                             /// settings of the bound-constrained QP of every pass
}

static assert(LeastSquaresSettings!double.sizeof == 128 && LeastSquaresSettings!double.qpSettings.offsetof == 104);
static assert(LeastSquaresSettings!float.sizeof == 68 && LeastSquaresSettings!float.qpSettings.offsetof == 56);

/// What a solve returns (reference lines 128-143); crosses the C ABI through the hidden result pointer.
struct LeastSquaresResult(T)
    if (is(T == double) || is(T == float))
{
    LeastSquaresStatus status = LeastSquaresStatus.numericError;
    uint iterations;          /// accepted steps
    uint fCalls;              /// residual evaluations (+n per finite-difference Jacobian)
    uint gCalls;              /// analytic Jacobian evaluations
    T residual = T.infinity;  /// sum of squares at the returned x
    T lambda = 0;             /// last damping value
}

static assert(LeastSquaresResult!double.sizeof == 32 && LeastSquaresResult!float.sizeof == 24);
static assert(LeastSquaresTask.sizeof == 16 && Slice!(double*).sizeof == 16);

// ------------------------------------------------------------------------------------------------------------
// C tier: prototypes of the symbols libmir_optim_amd.so exports (the reference DEFINES them at lines 637-799).
// Attributes are the reference's, so existing callers compile unchanged.
// ------------------------------------------------------------------------------------------------------------
extern(C) @system nothrow @nogc pure
{
    size_t mir_least_squares_work_length(size_t m, size_t n) @safe;
    size_t mir_least_squares_iwork_length(size_t m, size_t n) @safe;
    immutable(char)* mir_least_squares_status_string(LeastSquaresStatus st) @trusted;

    LeastSquaresResult!double mir_optimize_least_squares_d(
        scope const ref LeastSquaresSettings!double settings, size_t m, size_t n,
        double* x, const(double)* l, const(double)* u,
        Slice!(double*) work, Slice!(lapackint*) iwork,
        scope void* fContext, scope LeastSquaresFunctionBetterC!double f,
        scope void* gContext = null, scope LeastSquaresJacobianBetterC!double g = null,
        scope void* tmContext = null, scope LeastSquaresThreadManagerBetterC tm = null);

    LeastSquaresResult!float mir_optimize_least_squares_s(
        scope const ref LeastSquaresSettings!float settings, size_t m, size_t n,
        float* x, const(float)* l, const(float)* u,
        Slice!(float*) work, Slice!(lapackint*) iwork,
        scope void* fContext, scope LeastSquaresFunctionBetterC!float f,
        scope void* gContext = null, scope LeastSquaresJacobianBetterC!float g = null,
        scope void* tmContext = null, scope LeastSquaresThreadManagerBetterC tm = null);

    void mir_least_squares_init_d(ref LeastSquaresSettings!double settings) @safe;
    void mir_least_squares_init_s(ref LeastSquaresSettings!float settings) @safe;
    void mir_least_squares_reset_d(ref LeastSquaresSettings!double settings) @safe;
    void mir_least_squares_reset_s(ref LeastSquaresSettings!float settings) @safe;
}

alias mir_optimize_least_squares(T : double) = mir_optimize_least_squares_d;
alias mir_optimize_least_squares(T : float) = mir_optimize_least_squares_s;
alias mir_least_squares_init(T : double) = mir_least_squares_init_d;
alias mir_least_squares_init(T : float) = mir_least_squares_init_s;
alias mir_least_squares_reset(T : double) = mir_least_squares_reset_d;
alias mir_least_squares_reset(T : float) = mir_least_squares_reset_s;

/// Text for a status (the library holds the strings; they are NUL-terminated literals there).
string leastSquaresStatusString(LeastSquaresStatus st) @trusted pure nothrow @nogc
{
    auto p = mir_least_squares_status_string(st);
    size_t len;
    while (p[len]) ++len;
    return p[0 .. len];
}

// ------------------------------------------------------------------------------------------------------------
// extern(D) tier: the precompiled entry points the template tier calls (reference lines 594-635).
// ------------------------------------------------------------------------------------------------------------
private struct Bridge(T)
{
    LeastSquaresFunction!T f;
    LeastSquaresJacobian!T g;
    LeastSquaresThreadManager tm;

    extern(C) static void callF(scope void* self, size_t m, size_t n, const(T)* x, T* y) @system nothrow @nogc pure
    {
        (cast(Bridge*) self).f(x[0 .. n].sliced, y[0 .. m].sliced);
    }

    extern(C) static void callG(scope void* self, size_t m, size_t n, const(T)* x, T* J) @system nothrow @nogc pure
    {
        (cast(Bridge*) self).g(x[0 .. n].sliced, J[0 .. m * n].sliced(m, n));
    }

    // the library hands over its task as (opaque task value, C function); the user's manager wants a D delegate:
    // a delegate to a member of a stack struct carries both without a closure allocation
    static struct TaskCall
    {
        LeastSquaresTask task;
        LeastSquaresTaskBetterC fn;
        void opCall(uint totalThreads, uint threadId, uint i) @trusted nothrow @nogc pure
        {
            fn(task, totalThreads, threadId, i);
        }
    }

    extern(C) static void callTM(scope void* self, uint count, scope const LeastSquaresTask task, scope LeastSquaresTaskBetterC fn) @system nothrow @nogc pure
    {
        auto call = TaskCall(cast() task, fn);
        (cast(Bridge*) self).tm(count, cast(LeastSquaresTask) &call.opCall);
    }
}

private LeastSquaresResult!T viaCTier(T)(
    scope const ref LeastSquaresSettings!T settings, size_t m,
    Slice!(T*) x, Slice!(const(T)*) l, Slice!(const(T)*) u,
    Slice!(T*) work, Slice!(lapackint*) iwork,
    scope LeastSquaresFunction!T f, scope LeastSquaresJacobian!T g, scope LeastSquaresThreadManager tm) @trusted nothrow @nogc pure
{
    assert(l.length == x.length && u.length == x.length);
    auto b = Bridge!T(f, g, tm);
    return mir_optimize_least_squares!T(settings, m, x.length, x.ptr, l.ptr, u.ptr, work, iwork,
        &b, &Bridge!T.callF,
        g is null ? null : &b, g is null ? null : &Bridge!T.callG,
        tm is null ? null : &b, tm is null ? null : &Bridge!T.callTM);
}

pragma(inline, false)
LeastSquaresResult!double optimizeLeastSquaresD(
    scope const ref LeastSquaresSettings!double settings, size_t m,
    Slice!(double*) x, Slice!(const(double)*) l, Slice!(const(double)*) u,
    Slice!(double*) work, Slice!(lapackint*) iwork,
    scope LeastSquaresFunction!double f, scope LeastSquaresJacobian!double g = null,
    scope LeastSquaresThreadManager tm = null) @trusted nothrow @nogc pure
{
    return viaCTier!double(settings, m, x, l, u, work, iwork, f, g, tm);
}

/// (the reference's float instantiation passes the literal 2 for `m`, line 629; the caller's `m` is used here)
pragma(inline, false)
LeastSquaresResult!float optimizeLeastSquaresS(
    scope const ref LeastSquaresSettings!float settings, size_t m,
    Slice!(float*) x, Slice!(const(float)*) l, Slice!(const(float)*) u,
    Slice!(float*) work, Slice!(lapackint*) iwork,
    scope LeastSquaresFunction!float f, scope LeastSquaresJacobian!float g = null,
    scope LeastSquaresThreadManager tm = null) @trusted nothrow @nogc pure
{
    return viaCTier!float(settings, m, x, l, u, work, iwork, f, g, tm);
}

alias optimizeLeastSquares(T : double) = optimizeLeastSquaresD;
alias optimizeLeastSquares(T : float) = optimizeLeastSquaresS;

// ------------------------------------------------------------------------------------------------------------
// Template tier (reference lines 165-215, 459-519): lambdas / functors in, nothrow result or exception out.
// ------------------------------------------------------------------------------------------------------------
private enum isNull(alias a) = is(typeof(a) == typeof(null));

/++
Nothrow tier. `f(x, y)`: n -> m residuals; optional `g(x, J)`: row-major m x n Jacobian; optional `tm(count, task)`:
thread manager for the finite-difference columns. As in the reference, `y` / `J` are zero-filled before the user
code runs. The workspaces are only sized for ABI compatibility: the library keeps its state in HBM and ignores them, so
no allocation is made here -- the slices carry the required LENGTHS and a null pointer.
+/
LeastSquaresResult!T optimizeLeastSquares(alias f, alias g = null, alias tm = null, T)(
    scope const ref LeastSquaresSettings!T settings, size_t m,
    Slice!(T*) x, Slice!(const(T)*) l, Slice!(const(T)*) u)
    if (is(T == double) || is(T == float))
{
    scope fD = delegate(Slice!(const(T)*) xs, Slice!(T*) ys) { ys[] = 0; f(xs, ys); };
    static if (isNull!g)
        enum LeastSquaresJacobian!T gD = null;
    else
        scope gD = delegate(Slice!(const(T)*) xs, Slice!(T*, 2) Js) { Js[] = 0; g(xs, Js); };
    static if (isNull!tm)
        enum LeastSquaresThreadManager tmD = null;
    else
        scope tmD = delegate(uint count, scope LeastSquaresTask task) { tm(count, task); };

    const n = x.length;
    auto work = Slice!(T*)([mir_least_squares_work_length(m, n)], null);
    auto iwork = Slice!(lapackint*)([mir_least_squares_iwork_length(m, n)], null);
    return optimizeLeastSquares!T(settings, m, x, l, u, work, iwork,
        fD.assumeLmAttributes, gD.assumeLmAttributes, tmD.assumeLmAttributes);
}

version (D_Exceptions)
{
    // pre-built immutable exceptions, one per negative status, so that `optimize` stays @nogc
    private static immutable Exception[8] lsExceptions = () {
        import std.traits: EnumMembers;
        Exception[8] e;
        size_t k;
        static foreach (st; EnumMembers!LeastSquaresStatus)
            static if (st < 0)
                e[k++] = new Exception("mir-optim Least Squares: " ~ statusText(st));
        return e;
    }();

    // compile-time copy of the library's strings (CTFE cannot call into the shared object)
    private string statusText(LeastSquaresStatus st) @safe pure nothrow @nogc
    {
        final switch (st) with (LeastSquaresStatus)
        {
            case furtherImprovement: return "The algorithm cann't improve the solution";
            case maxIterations: return "Maximum number of iterations reached";
            case xConverged: return "X converged";
            case gConverged: return "Jacobian converged";
            case fConverged: return "Residual is small enough";
            case badBounds: return "Initial guess must be within bounds.";
            case badGuess: return "Initial guess must be an array of finite numbers.";
            case badMinStepQuality: return "0 <= minStepQuality < 1 must hold.";
            case badGoodStepQuality: return "0 < goodStepQuality <= 1 must hold.";
            case badStepQuality: return "minStepQuality < goodStepQuality must hold.";
            case badLambdaParams: return "1 <= lambdaIncrease && lambdaIncrease <= T.max.sqrt and T.min_normal.sqrt <= lambdaDecrease && lambdaDecrease <= 1 must hold.";
            case numericError: return "Numeric Error";
        }
    }

    private void throwFor(LeastSquaresStatus st) @trusted pure
    {
        // EnumMembers order: maxIterations (-1) first, then badBounds (-32) .. numericError (-26)
        throw cast() lsExceptions[st == LeastSquaresStatus.maxIterations ? 0 : st + 33];
    }
}

/// Throwing tier: a negative status becomes an exception (reference lines 165-181).
LeastSquaresResult!T optimize(alias f, alias g = null, alias tm = null, T)(
    scope const ref LeastSquaresSettings!T settings, size_t m,
    Slice!(T*) x, Slice!(const(T)*) l, Slice!(const(T)*) u)
    if (is(T == double) || is(T == float))
{
    auto ret = optimizeLeastSquares!(f, g, tm, T)(settings, m, x, l, u);
    version (D_Exceptions)
        if (ret.status < 0)
            throwFor(ret.status);
    return ret;
}

/// Task-pool overload (reference lines 184-215): finite-difference columns on `taskPool.parallel`.
LeastSquaresResult!T optimize(alias f, TaskPool, T)(
    scope const ref LeastSquaresSettings!T settings, size_t m,
    Slice!(T*) x, Slice!(const(T)*) l, Slice!(const(T)*) u, TaskPool taskPool)
    if (is(T == double) || is(T == float))
{
    auto manager = delegate(uint count, scope LeastSquaresTask task)
    {
        import mir.ndslice.topology: iota;
        const total = cast(uint) taskPool.size;
        foreach (i; taskPool.parallel(count.iota!uint))
            task(total, total <= 1 ? 0 : cast(uint)(taskPool.workerIndex - 1), i);
    };
    auto ret = optimizeLeastSquares!(f, null, manager, T)(settings, m, x, l, u);
    version (D_Exceptions)
        if (ret.status < 0)
            throwFor(ret.status);
    return ret;
}

// The precompiled tier is `@safe nothrow @nogc pure`; user lambdas are trusted to be (the reference does the same).
private auto assumeLmAttributes(D)(scope return D dg) @trusted
{
    import std.traits: FunctionAttribute, SetFunctionAttributes, functionAttributes, functionLinkage;
    static if (is(D == typeof(null)))
        return dg;
    else
    {
        enum attrs = (functionAttributes!D & ~FunctionAttribute.system) | FunctionAttribute.pure_ | FunctionAttribute.nothrow_
            | FunctionAttribute.nogc | FunctionAttribute.safe;
        return cast(SetFunctionAttributes!(D, functionLinkage!D, attrs)) dg;
    }
}
