"""C-ABI library (loads on a CPU-only machine, no compute calls) and the config layer."""
import ctypes as C
import os
import re
import sys
import subprocess

import numpy as np
import pytest

from options_in_simulation_amd import _abi, build_config, default_task_cfg
from options_in_simulation_amd.config import VARIANTS, composite_body

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(path):
    with open(path) as f:
        return f.read()


@pytest.fixture(scope="module")
def lib():
    from options_in_simulation_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _native.load()


def test_library_exports_every_symbol_of_the_header(lib):
    header = _read(os.path.join(ROOT, "include", "ois.h"))
    declared = set(re.findall(r"\b(ois_[a-z0-9_]+)\s*\(", header))
    declared -= {"ois_env"}
    assert declared, "no declarations parsed"
    from options_in_simulation_amd import _native
    assert declared == set(_native.SYMBOLS), "python binding list and include/ois.h disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (ois_[a-z0-9_]+)", nm))
    assert declared <= exported, f"missing exports: {declared - exported}"
    assert lib.ois_abi_version() == _abi.ABI_VERSION


def test_struct_layout_matches_the_library(lib):
    c = _abi.OisConfig()
    assert lib.ois_config_default(C.byref(c)) == 0
    assert c.struct_bytes == C.sizeof(_abi.OisConfig)  # sizeof(ois_config) as compiled by hipcc
    py = build_config(variant="KAMMA_medium", num_envs=1)
    for name, _ in _abi.OisConfig._fields_:
        a, b = getattr(c, name), getattr(py, name)
        a, b = (list(a), list(b)) if hasattr(a, "__len__") else (a, b)
        assert a == b, f"ois_config_default and build_config disagree on {name}: {a} vs {b}"


def test_workspace_size_and_config_validation(lib):
    c = build_config(variant="KAMMA_medium", num_envs=65536)
    nbytes = lib.ois_workspace_bytes(C.byref(c))
    state = 15 * 65536 * 16 + 65536 * (24 + 4 + 1 + 1)
    assert state <= nbytes < state + 4 * 1024 * 1024
    for field, bad in (("num_envs", 0), ("action_dim", 3), ("solver_iterations", 0), ("sim_dt", 0.0),
                       ("body_mass", -1.0), ("struct_bytes", 12), ("cmd_sampler", 9),
                       # the explicit joint-resistance step diverges at (c1 + c2) dt / Is >= 2; a negative depenetration
                       # velocity would push separating contacts (SPEC 3.3, 3.4)
                       ("wheel_limit_damping", 2.0), ("wheel_damping", 2.5), ("max_depenetration_velocity", -1.0),
                       ("friction", -0.1), ("friction", float("nan"))):
        c = build_config(variant="KAMMA_medium", num_envs=8)
        setattr(c, field, bad)
        assert lib.ois_workspace_bytes(C.byref(c)) == 0, field
        assert lib.ois_last_error().decode() != ""
        if field in ("wheel_limit_damping", "wheel_damping", "max_depenetration_velocity", "friction"):   # the oracle refuses the same
            from oracle.binding import lib as orc
            h = C.c_void_p()
            assert orc().orc_create(C.byref(c), C.byref(h)) != 0, field
    c = build_config(variant="KAMMA_medium", num_envs=8, wheel_limit_damping=1.9, max_depenetration_velocity=0.0)
    assert lib.ois_workspace_bytes(C.byref(c)) > 0


def test_null_and_order_errors_without_touching_a_gpu(lib):
    assert lib.ois_step(None, None, None) == _abi.EINVAL
    assert lib.ois_reset(None, 0, None) == _abi.EINVAL
    assert lib.ois_buffer(None, 0, None, None) == _abi.EINVAL
    assert lib.ois_workspace_bytes(None) == 0
    assert lib.ois_state_bytes(None) == 0
    h = C.c_void_p()
    c = build_config(variant="KAMMA_medium", num_envs=8)
    assert lib.ois_create(C.byref(c), 0, None, 0, C.byref(h)) == _abi.EINVAL  # null workspace, checked before HIP


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from options_in_simulation_amd.vecenv import NativeEnv, VecEnv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NativeEnv(build_config(variant="KAMMA_medium", num_envs=4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VecEnv(num_envs=4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "options_in_simulation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                txt = _read(os.path.join(dirpath, f))
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f      # python import
                assert not re.search(r'#\s*include\s+["<][^">]*oracle', txt), f          # C include
                assert "libois_oracle" not in txt and not re.search(r"\borc_[a-z_]+\s*\(", txt), f  # link / call


def test_task_cfg_surface_and_variants():
    cfg = default_task_cfg()
    env = cfg["task"]["env"]
    assert env["control"]["stiffness"] == 0.035 and env["randomCommandVelocityRanges"]["linear_x"] == [1.0, 2.0]
    c = build_config(cfg, variant="KAMMA_medium", num_envs=7, seed=9)
    assert (c.num_envs, c.seed, c.decimation, c.extra_substeps, c.max_episode_length) == (7, 9, 10, 10, 100)
    assert c.policy_dt == np.float32(0.1) and c.Kp == np.float32(0.035) and c.max_sim_steps == 500000.0
    raw = build_config(default_task_cfg(script_overrides=False), variant="KAMMA", num_envs=1)
    assert raw.Kp == 1.0 and raw.cmd_x_lo == raw.cmd_x_hi == 1.5 and raw.max_sim_steps == 400000.0
    assert set(VARIANTS) >= {"KAMMA", "KAMMA_medium", "KAMMA_g11_g21", "KAMMA_g1d_g21", "KAMMA_curr1", "KAMMA_curr2", "KAMMA_eval",
                             "KA_DDPG_4D", "KA_DDPG_2D", "KA_DDPG_1D"}
    e = build_config(variant="KAMMA_eval", num_envs=1)
    assert (e.gamma1_mode, e.gamma1_const, e.cmd_sampler, e.eval_sums) == (_abi.GAMMA_CONST, 0.0, _abi.CMD_EVAL, 1)
    d = build_config(variant="KA_DDPG_2D", num_envs=1)
    assert (d.action_dim, d.mix_mode, d.guiding_mode) == (2, _abi.MIX_BLEND, _abi.GUIDE_POLICY)
    with pytest.raises(ValueError):
        build_config(variant="nope")
    with pytest.raises(TypeError):
        build_config(variant="KAMMA", not_a_field=1)
    # patched keys flow through (what train_*.py:63-91 do)
    cfg["task"]["env"]["control"]["stiffness"] = 0.05
    cfg["task"]["sim"]["default_physics_material"]["static_friction"] = 0.6   # the key the train scripts patch (:88)
    cfg["task"]["env"]["learn"]["episodeLength_s"] = 5.0
    p = build_config(cfg, variant="KAMMA", num_envs=1)
    assert p.Kp == np.float32(0.05) and p.friction == np.float32(0.6) and p.max_episode_length == 50


def test_composite_body_from_urdf_numbers():
    m, com, (ixx, iyy, izz) = composite_body()
    assert m == 25.0  # = vehicle_mass KAMMA.py:174
    assert com == pytest.approx(4 * 0.056 / 25)
    assert ixx == pytest.approx(0.70054, abs=1e-4) and iyy == pytest.approx(0.94864, abs=1e-4) and izz == pytest.approx(1.4681, abs=1e-4)


def test_no_kernel_reads_the_dispatch_packet(tmp_path):
    """The AQL dispatch packet lives in host memory: a kernel that reads it (e.g. because the
    compiler promoted a run-time-indexed private array to LDS and linearises the thread id from
    the packet's workgroup size) pays ~13 us per launch on MI355X (tools/lds_launch_probe.hip,
    DESIGN.md section 4). No kernel of the library may enable the dispatch-pointer SGPRs; scratch is
    only allowed (<= 256 B per lane) in step_kernel_w2, the plane shell deliberately held to 256
    registers so that two waves fit a SIMD (launched only above one wave per SIMD of arenas)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "options_in_simulation_amd", "csrc")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
             "-Wno-unused-function", "-Wno-pass-failed", "-S", "--cuda-device-only"]
    for src in ("ois_kernels.hip", "ois_gae.hip"):
        out = tmp_path / (src + ".s")
        subprocess.run([hipcc, *flags, "-o", str(out), os.path.join(csrc, src)], check=True, capture_output=True)
        text = out.read_text()
        kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S)
        assert kernels
        for name, body in kernels:
            assert ".amdhsa_user_sgpr_dispatch_ptr 0" in body, f"{name} reads the dispatch packet"
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
            held = "step_kernel_w2" in name  # held to 256 registers on purpose: cold per-step state spills
            # elsewhere no instruction may touch scratch; the backend may still leave a dead spill
            # slot in the descriptor (36 B in step_kernel<1,true,4>: no access, no measured cost)
            code = text[text.index(name + ":"):text.index(".amdhsa_kernel " + name)]
            touches = re.findall(r"^\s+(scratch_\w+|buffer_(?:load|store)\w*)", code, flags=re.M)
            assert held or not touches, f"{name} accesses scratch: {sorted(set(touches))}"
            assert scratch <= (256 if held else 64), f"{name} uses {scratch} B of scratch per lane"
            if "step_kernel" in name:
                assert ".amdhsa_group_segment_fixed_size 0" in body, f"{name}: a private array was promoted to LDS"


def test_new_entry_points_reject_bad_arguments_without_a_gpu(lib):
    """replay / gae / terrain entry points: argument checks come before any HIP call"""
    assert lib.ois_replay_bytes(None, 4) == 0
    assert lib.ois_replay_attach(None, 4, None, 0) == _abi.EINVAL
    assert lib.ois_replay_sample(None, None, 1, None, None, None, None, None, None) == _abi.EINVAL
    i, f = C.c_int(), C.c_int()
    assert lib.ois_replay_cursor(None, C.byref(i), C.byref(f)) == _abi.EINVAL
    assert lib.ois_terrain_attach(None, None, None, None, None, None, None) == _abi.EINVAL
    assert lib.ois_height_scan(None, None, None) == _abi.EINVAL
    assert lib.ois_gae_scratch_bytes(0) == 0 and lib.ois_gae_scratch_bytes(65536) == (2 * 1024 + 1) * 8
    one = C.c_void_p(8)  # never dereferenced: every call below fails its checks first
    assert lib.ois_gae(7, 4, 4, one, one, one, one, 0.99, 0.95, 0, one, one, None, 0, None) == _abi.EINVAL  # unknown mode
    assert lib.ois_gae(0, 0, 4, one, one, one, one, 0.99, 0.95, 0, one, one, None, 0, None) == _abi.EINVAL  # T = 0
    assert lib.ois_gae(0, 4, 4, None, one, one, one, 0.99, 0.95, 0, one, one, None, 0, None) == _abi.EINVAL  # null tensor
    assert lib.ois_gae(0, 4, 4, one, one, one, one, 0.99, 0.95, 1, one, one, None, 0, None) == _abi.ENOMEM  # normalize without scratch
    assert b"scratch" in lib.ois_last_error()


def test_algorithmic_flop_count_is_the_one_bench_quotes():
    """docs/SPEC.md 3.11 = tools/spec_flops.py = bench.py's flop_algorithmic_per_env_step: a figure counted from the
    model, not from any kernel's instructions"""
    import re
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import spec_flops
    assert spec_flops.flop_per_substep() == 1653 and spec_flops.flop_per_env_step() == 20 * 1653 + 171
    assert bench.algorithmic_flop_per_env_step() == float(spec_flops.flop_per_env_step())
    spec = open(os.path.join(ROOT, "docs", "SPEC.md")).read()
    assert f"**{spec_flops.flop_per_substep()}**" in spec and f"**{spec_flops.flop_per_env_step()}**" in spec
    rows = re.findall(r"^\| 3\.\d \|.*\| (\d+) \| (\d+) \| (\d+) \|$", spec, flags=re.M)
    assert len(rows) == len(spec_flops.SUBSTEP) and all(int(a) * int(b) == int(c) for a, b, c in rows)
    # the executed figure is keyed on the launched instantiation and quoted only on this checkout's kernel sources
    flop, valu, prov = bench.executed_flop_per_env_step("step_kernel<1,1,4,1>", 20)
    # profiles/r4_isa_census.json is regenerated (tools/isa_census.py --json, no GPU needed) whenever the kernel sources change
    assert prov["status"].startswith("quoted"), prov
    assert valu == 20 * 595 and 0.8 < flop / bench.algorithmic_flop_per_env_step() < 1.0
    flop2, _, prov2 = bench.executed_flop_per_env_step("step_kernel<9,9,9,9>", 20)
    assert flop2 == bench.FLOP_PER_ENV_STEP and prov2["status"].startswith("not quoted")


def test_census_ids_are_the_ids_ois_kernel_info_reports():
    """tools/isa_census.py names a kernel by its template arguments without the trailing FOLD flag (`+fold` for the continuation
    shells): the string ois_kernel_info reports for the shell an ois_step launches, which is what bench.py looks up."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_census
    assert isa_census.kernel_id("_ZN12_GLOBAL__N_111step_kernelILi1ELb1ELi4ELb1ELb0EEEvN3ois9DevConstsE") == "step_kernel<1,1,4,1>"
    assert isa_census.kernel_id("_ZN12_GLOBAL__N_111step_kernelILi2ELb1ELi4ELb0ELb1EEEvN3ois9DevConstsE") == "step_kernel<2,1,4,0>+fold"
    assert isa_census.kernel_id("_ZN12_GLOBAL__N_114step_kernel_w2ILb1ELi4ELb1ELb0EEEvN3ois9DevConstsE") == "step_kernel_w2<1,4,1>"
    assert isa_census.kernel_id("_ZN12_GLOBAL__N_119step_kernel_terrainILb1ELi4ELb0EEEvN3ois9DevConstsE") == "step_kernel_terrain<1,4>"
    d = json.load(open(os.path.join(ROOT, "profiles", "r4_isa_census.json")))["kernels"]
    for kid in ("step_kernel<1,1,4,1>", "step_kernel<1,1,4,0>", "step_kernel<2,1,4,1>", "step_kernel_w2<1,4,1>", "step_kernel_terrain<1,4>",
                "step_kernel<1,1,4,1>+fold"):
        assert kid in d and d[kid]["scratch_bytes"] <= (256 if "w2" in kid else 64), kid
    assert d["step_kernel<1,1,4,0>"]["substep_valu"] - d["step_kernel<1,1,4,1>"]["substep_valu"] == 12   # the flat-plane shells


def test_observed_components_are_the_references_python_double_arithmetic():
    """task.observed_components (KAMMA.py:840-867) is rebuilt on the host from the 32-float env-0 block after EVERY step of a
    trainer that logs, in plain Python floats since round 4 (it used to cost 20 us of numpy scalars per step). Same values as the
    float64 numpy evaluation of the reference's expression `100.0 * (1 - gamma2) * comp / rew`, left to right, incl. what the
    reference's ZeroDivisionError case becomes here (inf / nan by the IEEE rules, both signs of zero)."""
    import numpy as np
    from options_in_simulation_amd.vecenv import observed_components_from_diag
    from options_in_simulation_amd.tracking import ENV0_TAGS
    rng = np.random.default_rng(0)
    cases = [rng.normal(size=32).astype(np.float32) for _ in range(50)]
    for rew in (0.0, -0.0, np.float32(1e-12), np.float32(np.nan)):
        for comp in (1.5, -2.0, 0.0, np.float32(np.inf)):
            d = rng.normal(size=32).astype(np.float32)
            d[27], d[20], d[26] = rew, comp, -comp
            cases.append(d)
    for d in cases:
        got = observed_components_from_diag(d)
        d64 = d.astype(np.float64)
        g2, rew = d64[19], d64[27]
        with np.errstate(divide="ignore", invalid="ignore"):
            want = {"env0_perc_r1": np.float64(100.0) * (1 - g2) * d64[20] / rew, "env0_perc_dense": np.float64(100.0) * (1 - g2) * d64[23] / rew,
                    "env0_perc_guiding": np.float64(100.0) * g2 * d64[26] / rew}
        for k, w in want.items():
            assert got[k] == float(w) or (np.isnan(got[k]) and np.isnan(w)), (k, got[k], w, d[27])
        assert got["env0_desired_v"] == float(d64[0]) and got["env0_episode_count"] == int(d64[8]) and isinstance(got["env0_episode_count"], int)
        assert set(ENV0_TAGS) <= set(got) and len(got) == 26   # the 26 keys the trainer reads (base_KA_DDPG.py:184-209)


def test_committed_profiles_were_taken_on_these_kernel_sources():
    """every profiles/r4_* artefact bench.py quotes names the sha256 of the kernel sources it was taken on; at a commit they
    are the checkout's (a kernel edit without `tools/profile_round.sh` + `tools/isa_census.py --json` shows up here, not in a
    silently stale roofline)"""
    import json
    import bench
    h = bench.kernel_source_hash()
    assert h in open(os.path.join(ROOT, "profiles", f"{bench.ROUND}_kernel_src_sha256.txt")).read()
    for name in (f"{bench.ROUND}_isa_census.json", f"{bench.ROUND}_pmc_sq.json", f"{bench.ROUND}_pmc_traffic.json"):
        assert json.load(open(os.path.join(ROOT, "profiles", name)))["kernel_src_sha256"] == h, name
