"""Exact integer sequence of the reference's ActiveSet test (tests/ActiveSetTest.cpp:60-133) replayed on the
oracle's restatement of src/internal/ActiveSet.cpp:45-168."""
I, L, U, E, LB, UB, FX = range(7)


def _check(o, h, status, active, nb):
    assert [o.lib.orc_as_status(h, i) for i in range(8)] == status
    import ctypes as ct

    c = (ct.c_int * 8)()
    o.lib.orc_as_counters(h, c)
    assert list(c) == list(nb)
    assert [o.lib.orc_as_at(h, k) for k in range(c[0])] == active


def test_activeset_sequence(oracle):
    o = oracle
    h = o.lib.orc_as_new(5, 3)
    try:
        _check(o, h, [I] * 8, [], (0, 0, 0, 0, 0, 0, 0, 0))
        o.lib.orc_as_activate(h, 3, E)
        _check(o, h, [I, I, I, E, I, I, I, I], [3], (1, 1, 0, 0, 0, 0, 0, 0))
        o.lib.orc_as_activate(h, 6, UB)
        _check(o, h, [I, I, I, E, I, I, UB, I], [3, 6], (2, 1, 0, 0, 0, 1, 0, 1))
        o.lib.orc_as_activate(h, 2, L)
        _check(o, h, [I, I, L, E, I, I, UB, I], [3, 6, 2], (3, 1, 1, 1, 0, 1, 0, 1))
        o.lib.orc_as_activate(h, 4, U)
        _check(o, h, [I, I, L, E, U, I, UB, I], [3, 6, 2, 4], (4, 1, 2, 1, 1, 1, 0, 1))
        o.lib.orc_as_deactivate(h, 1)
        _check(o, h, [I, I, L, E, U, I, I, I], [3, 2, 4], (3, 1, 2, 1, 1, 0, 0, 0))
        o.lib.orc_as_activate(h, 7, LB)
        _check(o, h, [I, I, L, E, U, I, I, LB], [3, 2, 4, 7], (4, 1, 2, 1, 1, 1, 1, 0))
        o.lib.orc_as_deactivate(h, 2)
        _check(o, h, [I, I, L, E, I, I, I, LB], [3, 2, 7], (3, 1, 1, 1, 0, 1, 1, 0))
        o.lib.orc_as_deactivate(h, 2)
        _check(o, h, [I, I, L, E, I, I, I, I], [3, 2], (2, 1, 1, 1, 0, 0, 0, 0))
        o.lib.orc_as_deactivate(h, 0)
        _check(o, h, [I, I, L, I, I, I, I, I], [2], (1, 0, 1, 1, 0, 0, 0, 0))
        o.lib.orc_as_deactivate(h, 0)
        _check(o, h, [I] * 8, [], (0, 0, 0, 0, 0, 0, 0, 0))
        o.lib.orc_as_activate(h, 5, FX)
        o.lib.orc_as_reset(h)
        _check(o, h, [I] * 8, [], (0, 0, 0, 0, 0, 0, 0, 0))
    finally:
        o.lib.orc_as_free(h)
