"""kept for the tests' imports: the CORA flow lives in dcora_amd/cora_flow.py (product control flow), its oracle backend
and the oracle RA-SLAM loop in oracle/flows.py (test infrastructure)"""
from dcora_amd.cora_flow import MIN_EIG_TOL, PARAMS, ProductBackend, cora  # noqa: F401
from oracle.flows import OracleBackend, oracle_ra_rbcd_loop  # noqa: F401
