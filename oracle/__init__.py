"""CPU oracle for the CDAE hot path — TEST INFRASTRUCTURE ONLY (see cdae_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .binding import Oracle, OracleConfig, MfOracle, MfConfig, build, eval_rec_list, eval_topn, OracleHeap  # noqa: F401
