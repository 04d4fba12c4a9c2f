import os, sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_engine_spec import run_case
for tol in (0.05, 0.1, 0.2):
    bad = []
    for name in ["xl_c10240_b10_bbd0", "xl_c10240_b10_bbd1", "xl_c25600_b10_bbd0", "xl_c8192_b10_bbd1", "xl_c8192_b5_bbd1", "xl_c10240_b1_bbd0"]:
        try:
            run_case(name, backend="native", score_tol=tol, kv_dtype="float16", ffn_dtype="float16")
        except AssertionError as e:
            bad.append(name)
    print("tol", tol, "failing:", bad)
