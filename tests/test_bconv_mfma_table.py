"""The matrix-core base conversion (rns_kernels.cuh: bconv_mfma_kernel) on the CPU: its operand table and the
kernel's recombination, with the MFMA instruction emulated, against sum_i y_i M[j][i] mod p_j in 128-bit integers
(tests/cpp/bconv_mfma_table_test.cpp).  The GPU parity tests of modup / moddown / key switch cover the kernel itself."""
import os
import subprocess

BIN = os.path.join(os.path.dirname(__file__), "cpp", "bconv_mfma_table_test")


def test_table_and_recombination_on_cpu():
    assert os.path.exists(BIN), "tests/cpp/bconv_mfma_table_test not built (run __graft_entry__.build())"
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
