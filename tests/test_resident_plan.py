"""CPU tier: the host side of the resident-J solver's device header (include/mir_optim_amd_resident.hpp) -- how a problem is laid
over the chip, and the -3 of a problem that does not fit (no GPU needed: resident_plan is host code)."""
import ctypes as C

from mir_optim_amd import api


def plan(model, m, cus=256):
    out4, out2 = (C.c_int * 4)(), (C.c_size_t * 2)()
    rc = api.workloads_lib().wl_resident_plan(C.c_int(model), C.c_size_t(m), C.c_int(cus), out4, out2)
    return rc, list(out4), list(out2)


def test_cfg2_fits_the_chip():
    rc, (grid, rows, groups, n), (lds, ws) = plan(0, 100000)
    assert rc == 0 and n == 16 and grid == 256 and rows == 391 and groups == 16
    assert grid * rows >= 100000 and lds <= 160 * 1024 - 2048 and ws < 1 << 20
    # slice: J (padded to 400 x 16) + 3 vectors + the 2-column row data
    assert lds >= 8 * (400 * 16 + 3 * 400 + 400 * 2)


def test_slices_and_groups():
    for m in (1, 63, 64, 65, 1000, 16383, 16384, 99999):
        rc, (grid, rows, groups, n), _ = plan(2, m)
        assert rc == 0 and 1 <= grid <= 256 and grid * rows >= m and (grid - 1) * rows < m     # no workgroup without rows
        assert groups == min(16, grid)
    rc, (grid, rows, groups, n), _ = plan(2, 5000, cus=7)
    assert rc == 0 and grid <= 7


def test_too_large_is_minus_three_and_unknown_model_minus_one():
    assert plan(0, 1000000)[0] == -3
    assert plan(1, 200000)[0] == -3
    assert plan(99, 1000)[0] == -1
