"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/gpt_hip.h declares
(no compute without a GPU), the product path fails loudly without a device, and the host logic
(data ingest, hyperparameter views, priors, pickling) behaves like the reference."""
import os
import pickle
import re
import subprocess
import sys
import warnings

import numpy as np
import pytest
import scipy.stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def g():
    warnings.simplefilter("ignore")
    import gptools_amd
    return gptools_amd


def test_library_exports_every_declared_symbol():
    from gptools_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gpt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpt_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gpt_version() >= 100


def test_no_cpu_fallback_in_product_package():
    """The shipped package must not import or execute the oracle."""
    pkg = os.path.join(ROOT, "gptools_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_hot_path_fails_loudly_without_gpu(g):
    from gptools_amd import _lib
    k = g.SquaredExponentialKernel(num_dim=1, initial_params=[1.0, 0.3], param_bounds=[(0, 10)] * 2)
    gp = g.GaussianProcess(k, X=np.linspace(0, 1, 8), y=np.zeros(8), err_y=0.1)
    with pytest.raises(_lib.GPTBackendError):
        gp.compute_K_L_alpha_ll()
    with pytest.raises(_lib.GPTBackendError):
        k(np.zeros((2, 1)), np.zeros((2, 1)), np.zeros((2, 1), int), np.zeros((2, 1), int))
    # update_hyperparameters swallows every exception into +inf like the reference (gaussian_process.py:1391-1406)
    assert gp.update_hyperparameters([1.0, 0.3]) == np.inf


def test_add_data_validation_and_layout(g):
    k = g.SquaredExponentialKernel(num_dim=2, initial_params=[1, 1, 1], param_bounds=[(0, 10)] * 3)
    gp = g.GaussianProcess(k)
    gp.add_data(np.zeros((4, 2)), np.arange(4.0), err_y=0.1)
    gp.add_data(np.ones((3, 2)), np.arange(3.0), err_y=[0.1, 0.2, 0.3], n=[[1, 0], [0, 1], [0, 0]])
    assert gp.X.shape == (7, 2) and gp.n.shape == (7, 2) and gp.y.shape == (7,)
    assert gp.n.dtype.kind == "i" and gp.n[4].tolist() == [1, 0]
    np.testing.assert_allclose(gp.err_y, [0.1] * 4 + [0.1, 0.2, 0.3])
    assert gp.K_up_to_date is False
    with pytest.raises(ValueError):
        gp.add_data(np.zeros((2, 3)), np.zeros(2))
    with pytest.raises(ValueError):
        gp.add_data(np.zeros((2, 2)), np.zeros(2), err_y=-1.0)
    with pytest.raises(ValueError):
        gp.add_data(np.zeros((2, 2)), np.zeros(2), n=-1)
    with pytest.raises(ValueError):
        gp.add_data(np.zeros((2, 2)), np.zeros(2), err_y=[0.1, 0.1, 0.1])
    with pytest.raises(ValueError):
        gp.add_data(np.zeros((2, 2)), np.zeros((2, 2)))
    # 1-D convenience: a (1, M) X is transposed (gaussian_process.py:439-440)
    k1 = g.SquaredExponentialKernel(num_dim=1, initial_params=[1, 1], param_bounds=[(0, 10)] * 2)
    gp1 = g.GaussianProcess(k1)
    gp1.add_data(np.linspace(0, 1, 5), np.zeros(5))
    gp1.add_data(0, 0, n=1)
    assert gp1.X.shape == (6, 1) and gp1.n[-1, 0] == 1
    # T handling: block-diagonal growth with identity for earlier untransformed data (gaussian_process.py:470-491)
    gp1.add_data(np.linspace(0, 1, 4), [1.0, 2.0], T=np.ones((2, 4)) / 4)
    assert gp1.T.shape == (8, 10)
    np.testing.assert_array_equal(gp1.T[:6, :6], np.eye(6))
    with pytest.raises(g.GPArgumentError):
        g.GaussianProcess(k1, X=np.zeros(3))
    with pytest.raises(TypeError):
        g.GaussianProcess("not a kernel")


def test_hyperparameter_views_and_bounds(g):
    k = g.SquaredExponentialKernel(num_dim=2, initial_params=[1.0, 2.0, 3.0], fixed_params=[False, True, False],
                                   param_bounds=[(0, 10), (0, 20), (0, 30)], enforce_bounds=True)
    nk = g.DiagonalNoiseKernel(num_dim=2, initial_noise=0.5, noise_bound=(0, 5))
    mu = g.ConstantMeanFunction(initial_params=[0.3])
    gp = g.GaussianProcess(k, noise_k=nk, mu=mu)
    assert gp.params[:] == [1.0, 2.0, 3.0, 0.5, 0.3]
    assert list(gp.free_params[:]) == [1.0, 3.0, 0.5, 0.3]
    assert [tuple(b) for b in gp.free_param_bounds[:]] == [(0, 10), (0, 30), (0, 5), (-1e3, 1e3)]
    assert list(~gp.fixed_params) == [True, False, True, True, True]
    k.set_hyperparams([50.0, -1.0])          # clamped by enforce_bounds (kernel/core.py:271-280)
    assert list(k.params) == [10.0, 2.0, 0.0]
    with pytest.raises(ValueError):
        k.set_hyperparams([1.0])
    gp.free_params = [4.0, 5.0, 0.6, 0.7]
    assert gp.params[:] == [4.0, 2.0, 5.0, 0.6, 0.7]
    gp.params[0] = 7.0                        # CombinedBounds writes through (utils.py:163-171)
    assert k.params[0] == 7.0
    with pytest.raises(g.GPArgumentError):
        g.SquaredExponentialKernel(num_dim=1, fixed_params=[True, True])
    with pytest.raises(ValueError):
        g.SquaredExponentialKernel(num_dim=0)
    assert g.SquaredExponentialKernel(num_dim=3, param_bounds=[(0, 1)] * 4).param_names.tolist() == \
        ["\\sigma_f", "l_1", "l_2", "l_3"]
    s = k + g.Matern52Kernel(num_dim=2, initial_params=[1, 1, 1], param_bounds=[(0, 1)] * 3)
    assert s.num_params == 6 and len(s.free_params) == 5


def test_priors_match_reference_formulas(g):
    hp = g.UniformJointPrior(0, 20) * g.GammaJointPriorAlt(1, 0.7)       # demo/demo.py:111
    th = [1.8849006111246833, 0.97760159723344708]
    b = (1 + np.sqrt(1 + 4 * 0.49)) / (2 * 0.49)
    a = 1 + b
    want = -np.log(20.0) + scipy.stats.gamma.logpdf(th[1], a, loc=0, scale=1.0 / b)
    assert abs(hp(th) - want) < 1e-13
    assert hp([21.0, 1.0]) == -np.inf
    assert hp(th, hyper_deriv=0) == 0.0
    assert abs(hp(th, hyper_deriv=1) - ((a - 1) / th[1] - b)) < 1e-13
    u = g.UniformJointPrior([(0, 2), (1, 3)])
    assert abs(u([1, 2]) - (-2 * np.log(2.0))) < 1e-15 and u([3, 2]) == -np.inf
    assert g.UniformJointPrior([0, 1], ub=[2, 3]).bounds == [(0, 2), (1, 3)]
    draws = hp.random_draw(size=7)
    assert draws.shape == (2, 7) and (draws[0] >= 0).all() and (draws[0] <= 20).all()
    nrm = g.NormalJointPrior([0.0], [2.0])
    assert abs(nrm([1.0]) - scipy.stats.norm.logpdf(1.0, 0, 2)) < 1e-14
    assert abs(nrm([1.0], hyper_deriv=0) - (-0.25)) < 1e-14


def test_demo_fixture_prior_value(g, golden):
    g6 = golden("g6_demo")
    gp = g.GaussianProcess(g.SquaredExponentialKernel(
        hyperprior=g.UniformJointPrior(0, 20) * g.GammaJointPriorAlt(1, 0.7)))
    gp.k.params[:] = g6["demo_params"]
    assert abs(gp.hyperprior(gp.params) - float(g6["prior_demo"])) < 1e-12


def test_mean_functions(g):
    mu = g.ConstantMeanFunction(initial_params=[2.5])
    X = np.random.RandomState(0).rand(6, 2)
    n = np.array([[0, 0], [1, 0], [0, 0], [0, 1], [0, 0], [2, 0]])
    np.testing.assert_array_equal(mu(X, n), [2.5, 0, 2.5, 0, 2.5, 0])
    np.testing.assert_array_equal(mu(X, n, hyper_deriv=0), [1, 0, 1, 0, 1, 0])
    lin = g.LinearMeanFunction(num_dim=2, initial_params=[2.0, -1.0, 0.5])
    np.testing.assert_allclose(lin(X, n), np.where(n.sum(1) == 0, 2 * X[:, 0] - X[:, 1] + 0.5,
                                                   np.where(n[:, 0] == 1, 2.0, np.where(n[:, 1] == 1, -1.0, 0.0))))


def test_product_kernel_leibniz_rule_against_reference(g, golden, oracle):
    """ProductKernel (ref: kernel/core.py:587-671) is host logic over per-factor pair evaluations: with the factors
    evaluated by the CPU oracle (test stand-ins for the GPU pair list) the product rule must reproduce the reference's
    outputs -- SE * SE with orders 0..2 per point and dimension, SE * Matern52 with first derivatives."""
    from conftest import assert_close
    from gptools_amd.kernel.core import Kernel, ProductKernel

    class OracleKernel(Kernel):
        def __init__(self, name, params):
            Kernel.__init__(self, num_dim=2, num_params=3, initial_params=list(params), param_bounds=[(0.0, 1e3)] * 3)
            self.name = name

        def __call__(self, Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False):
            return oracle.kpairs(self.name, self.params, Xi, Xj, ni, nj, hyper_deriv=hyper_deriv, symmetric=symmetric)

    G = golden("g9_product")
    k = OracleKernel("se", G["sese_p1"]) * OracleKernel("se", G["sese_p2"])
    assert isinstance(k, ProductKernel) and k.num_params == 6
    assert_close(k(G["sese_Xi"], G["sese_Xj"], G["sese_ni"], G["sese_nj"]), G["sese_k"], rtol=1e-11, msg="SE * SE")
    k = OracleKernel("se", G["sese_p1"]) * OracleKernel("m52", G["sese_p2"])
    assert_close(k(G["sese_Xi"], G["sese_Xj"], G["sem_ni"], G["sem_nj"]), G["sem_k"], rtol=1e-11, msg="SE * M52")
    with pytest.raises(NotImplementedError):
        k(G["sese_Xi"], G["sese_Xj"], G["sem_ni"], G["sem_nj"], hyper_deriv=0)


def test_gp_pickles_without_device_state(g):
    k = g.Matern52Kernel(num_dim=2, initial_params=[1, 0.5, 0.5], param_bounds=[(0, 10)] * 3)
    gp = g.GaussianProcess(k, X=np.random.rand(5, 2), y=np.random.rand(5), err_y=0.1)
    gp2 = pickle.loads(pickle.dumps(gp))
    assert gp2._ctx_obj is None and gp2.K_up_to_date is False
    np.testing.assert_array_equal(gp2.X, gp.X)
    assert gp2.k.params.tolist() == gp.k.params.tolist()


def test_bench_watchdog_prints_the_complete_line_and_exits():
    """bench.py's partitioned path measures the whole-panel schedule first and runs everything newer under a timer: if a
    later leg hangs, rank 0 prints the line that is already complete (flagged) and every rank's process ends with
    status 3 -- non-zero, so that the launcher sees the hang (bench.Watchdog)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog(int(sys.argv[1])); wd.line = {'metric': 'm', 'value': 1.0}; wd.phase = 'tuning pass, x'\n"
            "wd.arm(0.3); time.sleep(30); print('not reached')\n" % root)
    for rank, expect_line in ((0, True), (1, False)):
        out = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 3 and "not reached" not in out.stdout
        if expect_line:
            line = json.loads(out.stdout.strip())
            assert line["value"] == 1.0 and "tuning pass, x" in line["watchdog"]
        else:
            assert out.stdout.strip() == ""
    code_ok = ("import sys; sys.path.insert(0, %r); import bench\n"
               "wd = bench.Watchdog(0); wd.line = {}; wd.arm(30.0); print(wd.finish(), wd.finish())\n" % root)
    out = subprocess.run([sys.executable, "-c", code_ok], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.split() == ["True", "False"]



def test_bench_parity_gate_reports_and_gates():
    """bench.parity_report: within tolerance -> ok; an ll off by 1e-6 relative or a predictive mean off by
    1e-4 -> not ok (bench.py then prints the line and exits non-zero)."""
    import bench
    ref = {"ll_data": -1234.5, "logdet_half": 987.0}
    m = np.linspace(0, 1, 8)
    rep, ok = bench.parity_report(-1234.5 * (1 + 1e-10), 987.0, ref, (m, 0.1 + m), (m + 1e-9, 0.1 + m))
    assert ok and rep["ok"] and rep["ll_rel_err_vs_cpu"] < 1e-8
    assert not bench.parity_report(-1234.5 * (1 + 1e-6), 987.0, ref)[1]
    assert not bench.parity_report(-1234.5, 987.0 * (1 - 1e-7), ref)[1]
    assert not bench.parity_report(-1234.5, 987.0, ref, (m, 0.1 + m), (m + 1e-4, 0.1 + m))[1]


def test_bench_self_launch_starts_the_ranks_and_relays_their_status():
    """`python bench.py --gpus N` with no launcher in the environment starts its N ranks itself (bench.self_launch): fresh
    children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the parent never imports torch, and a failing rank's
    status comes back.  Without a GPU every rank stops at "needs an MI355X" -- which is exactly what is relayed here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(GPT_BENCH_GRACE_S="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 1, (out.returncode, out.stderr[-500:])
    assert out.stderr.count("needs an MI355X") == 2 and out.stdout.strip() == ""
    # the rank environment a child sees (the hook prints it instead of running)
    code = ("import os, sys; sys.argv = ['bench.py', '--gpus', '3']; sys.path.insert(0, %r)\n"
            "import bench, subprocess\n"
            "seen = []\n"
            "class P(object):\n"
            "    def __init__(self, cmd, env=None, **kw):\n"
            "        seen.append((env['RANK'], env['LOCAL_RANK'], env['WORLD_SIZE'], env['MASTER_ADDR'], cmd[1]))\n"
            "        self.stdout = []\n"
            "    def poll(self): return 0\n"
            "subprocess.Popen = P\n"
            "try: bench.self_launch(3)\n"
            "except SystemExit as e: print('exit', e.code)\n"
            "print(seen); print('torch' in sys.modules)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    lines = out.stdout.splitlines()
    assert lines[0] == "exit 0" and lines[2] == "False"
    seen = eval(lines[1])
    assert [s[:4] for s in seen] == [(str(r), str(r), "3", "127.0.0.1") for r in range(3)]
    assert all(s[4].endswith("bench.py") for s in seen)


def test_oracle_restatement_is_clean_under_asan_and_ubsan():
    """SURVEY.md section 5 (sanitizers on the CPU build): scripts/asan_cpu.sh builds oracle/gpt_oracle.c with
    -fsanitize=address,undefined and runs the golden-vector suite against it with the runtime preloaded; any report aborts.
    Skipped where gcc ships no libasan."""
    import shutil
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not shutil.which("gcc") or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no AddressSanitizer runtime here")
    p = subprocess.run([os.path.join(ROOT, "scripts", "asan_cpu.sh")], capture_output=True, text=True, timeout=900)
    if p.returncode == 77:
        pytest.skip("asan runtime not found")
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
    # the sanitised build really was the library under test
    q = subprocess.run([sys.executable, "-c",
                        "import os, sys; sys.path.insert(0, %r)\n"
                        "from oracle import oracle as O\n"
                        "O.lib(); print([l.split()[-1] for l in open('/proc/self/maps') if 'libgpt_oracle' in l][0])" % ROOT],
                       capture_output=True, text=True,
                       env=dict(os.environ, GPT_ORACLE_LIB=os.path.join(ROOT, "oracle", "libgpt_oracle_asan.so"),
                                LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0"))
    assert q.stdout.strip().endswith("libgpt_oracle_asan.so"), (q.stdout, q.stderr[-500:])


def test_every_context_option_is_documented_in_the_header():
    """Every key gpt_ctx_set_option accepts appears (quoted) in the option list of include/gpt_hip.h -- the C ABI's only documentation."""
    import re
    keys = re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', open(os.path.join(ROOT, "gptools_amd", "csrc", "api_context.inc")).read())
    hdr = open(os.path.join(ROOT, "include", "gpt_hip.h")).read()
    assert len(keys) >= 30
    missing = [k for k in keys if '"%s"' % k not in hdr]
    assert not missing, missing
