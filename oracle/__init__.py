"""oracle -- CPU restatement of the BspAtom hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (bspatom_amd) never does.  See oracle/bsp_oracle.c for the
reference file:line citations and oracle/ref/build_ref.sh for the compiled reference.
"""
from .oracle import *  # noqa: F401,F403
