"""Fortran NAMELIST reader for the bsp_0.inp input contract (host side, pure Python).

Mirrors what `READ(5,VARS_BSP)`, `READ(5,VARS_TISE)`, `READ(5,VARS_FIELD)` accept in
READ_INPUTS (reference ReadInputs.f90:15-21,37,85,184): groups in that fixed order, free text
and `!` comment lines between groups, `&NAME ... &end` or `/` terminators, whitespace- or
comma-separated KEY=value pairs that may span lines, case-insensitive names, D-exponent reals.
An unknown key is an error, as it is at run time in Fortran.
"""
import re

VARS_BSP = ("kind_grid", "ra", "rb", "rmax", "k", "ka", "nfun", "kind_bc1", "kind_bc2", "nfib")
VARS_TISE = ("n0_ini", "l_ini", "m_ini", "l_fin", "lmax", "emax_fin", "zatom", "kind_pot",
             "kind_egr", "kind_nlm")
VARS_FIELD = ("kind_pi", "kind_scp", "kind_td", "kind_env", "kind_rk", "kind_vec", "a0", "w0", "eph",
              "ncyc", "eph2", "ncyc2", "moam", "mph", "i0", "i01", "b0", "afocus", "nepts", "nthpts",
              "nphpts", "eref", "bx", "b0z", "a01", "t_delay", "a0x", "a0y", "a0z")
GROUPS = (("vars_bsp", VARS_BSP), ("vars_tise", VARS_TISE), ("vars_field", VARS_FIELD))
_INT_KEYS = {"kind_grid", "k", "ka", "nfun", "kind_bc1", "kind_bc2", "nfib", "n0_ini", "l_ini", "m_ini",
             "l_fin", "lmax", "kind_pot", "kind_egr", "kind_nlm", "kind_pi", "kind_scp", "kind_td",
             "kind_env", "kind_rk", "kind_vec", "ncyc", "ncyc2", "moam", "mph", "nepts", "nthpts", "nphpts"}


class NamelistError(ValueError):
    pass


def _value(key, tok):
    t = tok.strip().rstrip(",")
    if key in _INT_KEYS:
        try:
            return int(t)
        except ValueError:
            raise NamelistError("bad integer for %s: %r" % (key, tok))
    try:
        return float(re.sub(r"[dD]", "e", t))
    except ValueError:
        raise NamelistError("bad real for %s: %r" % (key, tok))


def read_namelists(text):
    """Return {'vars_bsp': {...}, 'vars_tise': {...}, 'vars_field': {...}} with lower-case keys."""
    out = {}
    pos = 0
    for gname, keys in GROUPS:
        # like a Fortran namelist READ: skip forward until '&gname' is found
        m = re.compile(r"&\s*" + gname + r"\b", re.IGNORECASE).search(text, pos)
        if m is None:
            raise NamelistError("namelist group &%s not found (groups must appear in the order "
                                "VARS_BSP, VARS_TISE, VARS_FIELD)" % gname.upper())
        end = re.compile(r"(&\s*end\b|/)", re.IGNORECASE).search(text, m.end())
        if end is None:
            raise NamelistError("namelist group &%s is not terminated" % gname.upper())
        body = text[m.end(): end.start()]
        body = re.sub(r"!.*", "", body)                      # in-group comments
        vals = {}
        for km in re.finditer(r"([A-Za-z_][A-Za-z0-9_]*)\s*=\s*([^\s,=]+)", body):
            key = km.group(1).lower()
            if key not in keys:
                raise NamelistError("unknown key %s in namelist %s" % (km.group(1), gname.upper()))
            vals[key] = _value(key, km.group(2))
        out[gname] = vals
        pos = end.end()
    return out
