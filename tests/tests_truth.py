"""Helpers shared by the truth-fixture tests: the oracle configuration of a golden input."""
import os
import oracle as orc
from bspatom_amd.namelist import read_namelists
from conftest import GOLDEN


def case_cfg(name):
    nl = read_namelists(open(os.path.join(GOLDEN, "inputs", name + ".inp")).read())
    kw = {}
    kw.update(nl["vars_bsp"]); kw.update(nl["vars_tise"])
    return orc.make_cfg(**kw)
