"""The deck of the reference's tests/test_ecl_output.cc on the device: the intensive quantities k_iq_update leaves behind give the
field / region pressures and fluids in place the reference expects (tests/test_oracle_ecl_output.py has the oracle's side), and equal
the oracle's records bit for bit."""
import numpy as np
import pytest

import helpers
import oracle_bind
from test_oracle_ecl_output import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reorder", [None, "line_coloring"])
def test_summary_deck_on_the_device(pkg, orc, reorder):
    case, fipnum = helpers.summary_deck_case(pkg)
    m = pkg.capi.HipModel(case, **({} if reorder is None else {"reorder": reorder}))
    m.set_state(case["pv"], case["meaning"])
    iq = m.iq()
    check(helpers.summary_from_iq(iq, case["volume"], fipnum))
    o = oracle_bind.OracleModel(orc, case)
    o.set_state(case["pv"], case["meaning"])
    assert np.array_equal(iq, o.iq())
    # and the storage term the assembly forms from them: at iteration 0 the residual of a closed system at rest in its own time level is
    # flux only; the Jacobian's diagonal carries d(storage)/d(primary variables) * V / dt - bit for bit the oracle's
    jm, rm = m.assemble(86400.0, 0)
    jo, ro = o.assemble(86400.0, 0)
    assert np.array_equal(jm, jo) and np.array_equal(rm, ro)
