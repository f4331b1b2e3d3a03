"""Shared problem definitions for the parity tests.

The first block restates the INPUTS of the reference's own unittests
(/root/reference/source/mir/optim/least_squares.d:217-434, boxcqp.d:382-402) as data; the
second block builds the synthetic workloads of SURVEY.md section 8d with the counter RNG
u(k) = (splitmix64(seed + k) >> 11) * 2^-53 (numpy implementation below; the C twin is
oracle/workloads_cpu.c:wlc_uniform and mir_optim_amd/csrc/workloads.hip).
"""
import numpy as np

INF = np.inf


def splitmix64_uniform(seed, count, offset=0):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(offset) + np.arange(count, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


# ---------------------------------------------------------------- reference unittests (inputs only)
def t1():  # LS:218-245
    def f(x, y):
        y[0] = x[0]; y[1] = 2 - x[1]

    def g(x, J):
        J[0, 0] = 1; J[0, 1] = 0; J[1, 0] = 0; J[1, 1] = -1
    return dict(f=f, g=g, m=2, x0=[100.0, 100.0], lower=None, upper=None, expect=[0.0, 2.0], tol=1e-8)


def rosenbrock_f(x, y):  # LS:261-265, LS:289-293
    y[0] = 10 * (x[1] - x[0] ** 2)
    y[1] = 1 - x[0]


def rosenbrock_g(x, J):  # LS:295-301
    J[0, 0] = -20 * x[0]; J[0, 1] = 10; J[1, 0] = -1; J[1, 1] = 0


def t2():  # LS:248-273 (finite differences)
    return dict(f=rosenbrock_f, g=None, m=2, x0=[-1.2, 1.0], lower=None, upper=None, expect=[1.0, 1.0], tol=1e-6)


def t3a():  # LS:276-317
    return dict(f=rosenbrock_f, g=rosenbrock_g, m=2, x0=[-1.2, 1.0], lower=None, upper=None, expect=[1.0, 1.0], tol=1e-8)


def t3b():  # LS:321-330
    return dict(f=rosenbrock_f, g=rosenbrock_g, m=2, x0=[150.0, 150.0], lower=[10.0, 10.0], upper=[200.0, 200.0],
                expect=[10.0, 100.0], tol=1e-5)


def t4(noise_seed=12345):  # LS:334-363; mir-random's N(0,1) stream is replaced by numpy's (tolerance 0.05 absorbs it)
    rng = np.random.default_rng(noise_seed)
    t = np.linspace(0.0, 10.0, 20)
    yd = 1.0 * np.exp(-t * 2.0) + 0.01 * rng.standard_normal(20)

    def f(p, y):
        y[:] = p[0] * np.exp(-t * p[1]) - yd
    return dict(f=f, g=None, m=20, x0=[0.5, 0.5], lower=None, upper=None, expect=[1.0, 2.0], tol=0.05, t=t, data=yd)


def t5(noise_seed=12345):  # LS:366-411
    rng = np.random.default_rng(noise_seed)
    t = np.arange(1.0, 101.0)
    yd = 10.0 * np.exp(-t / 10.0) + 10.0 + 0.1 * rng.standard_normal(100)

    def f(p, y):
        y[:] = p[0] * np.exp(-t / p[1]) + p[2] - yd
    a = dict(f=f, g=None, m=100, x0=[15.0, 15.0, 15.0], lower=[5.0, 11.0, 5.0], upper=None, t=t, data=yd)
    b = dict(f=f, g=None, m=100, x0=[5.0, 5.0, 5.0], lower=None, upper=[15.0, 9.0, 15.0], t=t, data=yd)
    return a, b


def t6():  # LS:414-434
    def f(x, y):
        y[0] = np.sqrt(1 - (x[0] ** 2 + x[1] ** 2))
    return dict(f=f, g=None, m=1, x0=[0.001, 0.0001], lower=[-0.5, -0.5], upper=[0.5, 0.5], expect=[0.5, 0.5], tol=1e-8)


def tq():  # QP:382-402
    P = np.array([[2.0, -1, 0], [-1.0, 2, -1], [0.0, -1, 2]])
    return dict(P=P, q=[3.0, -7, 5], l=[-100.0, -2, 1], u=[100.0, 2, 1], expect=[-0.5, 2.0, 1.0])


# ---------------------------------------------------------------- synthetic workloads
def tanh_linear(m, n, row_offset=0, m_total=None, noise=1e-3):
    """cfg 3 / cfg 4 family: r_i(x) = tanh(a_i . x) - b_i, seeds 10..13. Rows [row_offset, row_offset+m)."""
    A = ((2 * splitmix64_uniform(10, m * n, row_offset * n) - 1) * np.sqrt(3.0 / n)).reshape(m, n)
    xs = 2 * splitmix64_uniform(11, n) - 1
    dots = np.zeros(m)
    for j in range(n):                      # fixed summation order: a row's value does not depend on the shard it is in
        dots += A[:, j] * xs[j]
    b = np.tanh(dots) + noise * (2 * splitmix64_uniform(12, m, row_offset) - 1)
    x0 = xs + 0.1 * (2 * splitmix64_uniform(13, n) - 1)
    return dict(A=np.ascontiguousarray(A), b=b, xstar=xs, x0=x0, m=m, n=n)


def gauss_sum(m, K=5, noise=1e-3):
    """cfg 2 family: sum_k a_k exp(-(t-c_k)^2/(2 w_k^2)) + b; n = 3K+1."""
    n = 3 * K + 1
    t = np.arange(m, dtype=np.float64) / (m - 1)
    a = np.array([1, .8, .6, .9, .7, .5, .4, .3])[:K]
    c = (np.arange(K) + 0.5) / K
    w = np.full(K, 0.2 / K)
    truth = np.concatenate([a, c, w, [0.1]])
    model = sum(a[k] * np.exp(-(t - c[k]) ** 2 / (2 * w[k] ** 2)) for k in range(K)) + 0.1
    data = model + noise * (2 * splitmix64_uniform(2, m) - 1)
    x0 = truth * (1 + 0.05 * (2 * splitmix64_uniform(3, n) - 1))
    lower = np.full(n, -INF)
    lower[2 * K:3 * K] = 1e-3
    upper = np.full(n, INF)
    return dict(t=t, data=data, truth=truth, x0=x0, lower=lower, upper=upper, m=m, n=n, K=K)


def cfg5_pad8(count, m=512, noise=0.01):
    """BASELINE cfg 5: `count` independent fp32 fits of m points, n = 8 (the exponential decay
    p0 exp(-t p1) + p2 padded to n = 8 with five terms linear in their parameters; per-problem seed = 100 + problem id).
    Returns (t[m], data[count, m], truth[count, 8], x0[count, 8]) in float32."""
    t = np.linspace(0.0, 4.0, m, dtype=np.float32)
    td = t.astype(np.float64)
    data = np.empty((count, m), dtype=np.float32)
    truth = np.empty((count, 8), dtype=np.float32)
    x0 = np.empty((count, 8), dtype=np.float32)
    basis = np.stack([np.sin(2 * td), np.cos(2 * td), np.sin(5 * td), np.cos(5 * td), td])
    for k in range(count):
        u = splitmix64_uniform(100 + k, m + 16)
        p = np.array([1.0 + u[0], 0.5 + 2.0 * u[1], 0.2 * u[2], 0.6 * u[3] - 0.3, 0.6 * u[4] - 0.3, 0.6 * u[5] - 0.3,
                      0.6 * u[6] - 0.3, 0.1 * u[7] - 0.05])
        truth[k] = p
        data[k] = p[0] * np.exp(-td * p[1]) + p[2] + p[3:] @ basis + noise * (2 * u[16:] - 1)
        x0[k] = p
        x0[k, :2] *= 1 + 0.2 * (2 * u[8:10] - 1)
        x0[k, 2:] += 0.1 * (2 * u[10:16] - 1)
    return t, data, truth, x0
