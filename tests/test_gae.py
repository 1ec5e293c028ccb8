"""Advantage estimation (include/ois.h ois_gae): oracle vs the torch-loop fixture of
src/my_agents/ppo.py:345-358 and src/my_agents/ppoc.py:325-336, and HIP kernel vs oracle.

Bar: the un-normalised recurrences are float32 operation for operation the reference's, so
returns and raw advantages are compared BIT-EXACT (fixture, oracle and kernel). The
normalised advantages go through a global mean / std whose summation order is the
library's own (torch's is not specified), hence rtol 1e-5 (north_star tolerance) + atol 1e-6.
"""
import os

import numpy as np
import pytest

from oracle import binding as ob

G = os.path.join(os.path.dirname(__file__), "golden")
RTOL, ATOL = 1e-5, 1e-6


def _fixture():
    d = np.load(os.path.join(G, "refmath_gae.npz"))
    return d, float(d["discount_factor"]), float(d["lambda_coefficient"])


def test_oracle_matches_torch_loop_fixture():
    d, g, l = _fixture()
    ret, adv = ob.gae(0, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, False)
    assert np.array_equal(adv, d["ppo_advantages_raw"]) and np.array_equal(ret, d["ppo_returns"])
    ret, adv = ob.gae(0, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, True)
    assert np.array_equal(ret, d["ppo_returns"])
    np.testing.assert_allclose(adv, d["ppo_advantages"], rtol=RTOL, atol=ATOL)
    ret, adv = ob.gae(1, d["rewards"], d["terminated"], d["values"], d["last_values"], g, l, False)
    assert np.array_equal(adv, d["ppoc_advantages"]) and np.array_equal(ret, d["ppoc_returns"])


def test_oracle_gae_properties():
    rng = np.random.default_rng(0)
    T, N = 13, 9
    r = rng.normal(size=(T, N)).astype(np.float32)
    v = rng.normal(size=(T, N)).astype(np.float32)
    lv = rng.normal(size=N).astype(np.float32)
    none = np.zeros((T, N), np.uint8)
    # lambda = 0: one-step TD error; lambda = 1, gamma = 1, no terminals: reward-to-go + bootstrap - v
    _, adv = ob.gae(0, r, none, v, lv, 0.9, 0.0, False)
    nxt = np.vstack([v[1:], lv[None]])
    np.testing.assert_allclose(adv, r + np.float32(0.9) * nxt - v, rtol=1e-6, atol=1e-6)
    ret, _ = ob.gae(0, r, none, v, lv, 1.0, 1.0, False)
    np.testing.assert_allclose(ret, np.cumsum(r[::-1], 0)[::-1] + lv, rtol=1e-5, atol=1e-5)
    # a terminal step cuts the recurrence: everything before it is independent of what follows
    term = none.copy(); term[6] = 1
    _, a1 = ob.gae(0, r, term, v, lv, 0.99, 0.95, False)
    r2, v2 = r.copy(), v.copy(); r2[7:] += 5; v2[7:] -= 3
    _, a2 = ob.gae(0, r2, term, v2, lv + 1, 0.99, 0.95, False)
    assert np.array_equal(a1[:7], a2[:7]) and not np.array_equal(a1[7:], a2[7:])
    # the two agents' expressions agree up to rounding; normalised advantages are standardised
    _, a3 = ob.gae(1, r, term, v, lv, 0.99, 0.95, False)
    np.testing.assert_allclose(a1, a3, rtol=1e-5, atol=1e-5)
    _, an = ob.gae(0, r, term, v, lv, 0.99, 0.95, True)
    assert abs(an.mean()) < 1e-6 and an.std(ddof=1) == pytest.approx(1.0, abs=1e-5)
    with pytest.raises(RuntimeError):
        ob.gae(2, r, term, v, lv, 0.99, 0.95, False)


# ------------------------------------------------------------------------------------- gpu
torch = pytest.importorskip("torch")


@pytest.mark.gpu
def test_kernel_matches_fixture_and_oracle_bit_exact():
    from options_in_simulation_amd import compute_advantages_and_returns, compute_gae
    d, g, l = _fixture()
    dev = {k: torch.from_numpy(d[k]).cuda() for k in ("rewards", "terminated", "values", "last_values")}
    ret, adv = compute_gae(dev["rewards"], dev["terminated"], dev["values"], dev["last_values"], g, l, normalize=False)
    assert ret.shape == dev["rewards"].shape
    assert np.array_equal(adv.cpu().numpy(), d["ppo_advantages_raw"]) and np.array_equal(ret.cpu().numpy(), d["ppo_returns"])
    ret, adv = compute_gae(rewards=dev["rewards"], dones=dev["terminated"], values=dev["values"],
                           next_values=dev["last_values"], discount_factor=g, lambda_coefficient=l)
    assert np.array_equal(ret.cpu().numpy(), d["ppo_returns"])
    np.testing.assert_allclose(adv.cpu().numpy(), d["ppo_advantages"], rtol=RTOL, atol=ATOL)
    adv, ret = compute_advantages_and_returns(dev["rewards"], dev["terminated"], dev["values"], dev["last_values"], g, l)
    assert np.array_equal(adv.cpu().numpy(), d["ppoc_advantages"]) and np.array_equal(ret.cpu().numpy(), d["ppoc_returns"])


@pytest.mark.gpu
@pytest.mark.parametrize("T,N", [(1, 1), (2, 1), (7, 300), (16, 65536), (16, 4099), (20, 65536), (32, 40000), (33, 5000), (96, 4099), (16, 200000),
                                 (70, 65536), (16, 262144), (17, 262145), (5, 1000003)])
def test_kernel_vs_oracle_sizes(T, N):
    """incl. the reference's rollouts of 16 and 20 steps at 65 536 arenas, a ragged last block, and every path of the launch logic
    (ois_gae.hip): scan in four-wave blocks up to 262 144 arenas and in one-wave blocks above; the normalise kernel summing the
    per-block moments itself (<= 2048 of them) and the separate moments kernel (262 145 and 1 000 003 arenas); four advantages per
    access from 2^22 entries on (70 x 65 536; exactly 2^22 = 16 x 262 144; a count that is not a multiple of four)"""
    from options_in_simulation_amd import compute_advantages_and_returns, compute_gae
    rng = np.random.default_rng(T * 1000 + N)
    r = rng.normal(size=(T, N, 1)).astype(np.float32)
    v = rng.normal(size=(T, N, 1)).astype(np.float32)
    lv = rng.normal(size=(N, 1)).astype(np.float32)
    term = rng.uniform(size=(T, N, 1)) < 0.05
    tr, tv, tl, tt = (torch.from_numpy(x).cuda() for x in (r, v, lv, term))
    for mode in (0, 1):
        o_ret, o_adv = ob.gae(mode, r, term, v, lv, 0.99, 0.95, False)
        if mode == 0:
            k_ret, k_adv = compute_gae(tr, tt, tv, tl, 0.99, 0.95, normalize=False)
        else:
            k_adv, k_ret = compute_advantages_and_returns(tr, tt, tv, tl, 0.99, 0.95)
        assert np.array_equal(k_adv.cpu().numpy(), o_adv), f"mode {mode}: advantages not bit-identical"
        assert np.array_equal(k_ret.cpu().numpy(), o_ret), f"mode {mode}: returns not bit-identical"
    if T * N > 1:
        _, o_adv = ob.gae(0, r, term, v, lv, 0.99, 0.95, True)
        _, k_adv = compute_gae(tr, tt, tv, tl, 0.99, 0.95)
        np.testing.assert_allclose(k_adv.cpu().numpy(), o_adv, rtol=RTOL, atol=ATOL)
        # against torch's own reductions on the device as a second opinion
        raw = torch.from_numpy(ob.gae(0, r, term, v, lv, 0.99, 0.95, False)[1]).cuda()
        ref = (raw - raw.mean()) / (raw.std() + 1e-8)
        np.testing.assert_allclose(k_adv.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_normalisation_with_an_output_that_is_not_16_byte_aligned():
    """the four-per-access normalisation needs a 16-byte aligned advantages array; any other (here: 4 bytes into an allocation)
    takes the element-wise form: same values"""
    import ctypes as C
    from options_in_simulation_amd import _native
    L = _native.load()
    T, N = 72, 65536   # 4.7 Mi entries: above the threshold of the vector form
    rng = np.random.default_rng(3)
    r = torch.from_numpy(rng.normal(size=(T, N)).astype(np.float32)).cuda()
    v = torch.from_numpy(rng.normal(size=(T, N)).astype(np.float32)).cuda()
    lv = torch.from_numpy(rng.normal(size=(N,)).astype(np.float32)).cuda()
    d = torch.from_numpy((rng.uniform(size=(T, N)) < 0.05).astype(np.uint8)).cuda()
    nb = L.ois_gae_scratch_bytes(N)
    scratch = torch.empty(nb // 8, dtype=torch.float64, device="cuda")
    out = {}
    for shift in (0, 1):
        ret = torch.empty(T * N + 4, dtype=torch.float32, device="cuda")
        adv = torch.empty(T * N + 4, dtype=torch.float32, device="cuda")
        rc = L.ois_gae(0, T, N, r.data_ptr(), d.data_ptr(), v.data_ptr(), lv.data_ptr(), 0.99, 0.95, 1, ret.data_ptr() + 4 * shift,
                       adv.data_ptr() + 4 * shift, scratch.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, L.ois_last_error()
        torch.cuda.synchronize()
        out[shift] = (ret[shift:shift + T * N].clone(), adv[shift:shift + T * N].clone())
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    _, o_adv = ob.gae(0, r.cpu().numpy()[..., None], d.cpu().numpy().astype(bool)[..., None], v.cpu().numpy()[..., None], lv.cpu().numpy()[..., None], 0.99, 0.95, True)
    np.testing.assert_allclose(out[1][1].cpu().numpy().reshape(T, N, 1), o_adv, rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
def test_gae_error_paths():
    from options_in_simulation_amd import compute_gae
    z = torch.zeros((4, 8, 1), device="cuda")
    with pytest.raises(ValueError):
        compute_gae(z, z.bool(), z[:3], z[0])
    with pytest.raises(ValueError):
        compute_gae(z.cpu(), z.bool().cpu(), z.cpu(), z[0].cpu())
