"""The ctypes stub printed in INTEGRATION.md is executed as written (only the input path, table size and row filter are
substituted) and its matrices are compared with the oracle -- the document cannot drift from the ABI."""
import os
import re

import numpy as np
import pytest

from oracle import oracle

from .conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_integration_stub_runs_and_matches_oracle():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "pg_ingest_fastq" in b)
    fq = os.path.join(GOLDEN, "tenx_clean.fq.gz")
    stub = stub.replace('"libpangaea_feat.so"', repr(os.path.join(ROOT, "pangaea_amd", "libpangaea_feat.so")))
    stub = stub.replace('b"reads.sorted.fastq"', repr(fq.encode()))
    stub = stub.replace("1 << 29", "1 << 16").replace("pg_table(2, 21, 29, 0,", "pg_table(2, 21, 16, 0,")
    stub = stub.replace("L.pg_reads_rows(h, 2000,", "L.pg_reads_rows(h, 1000,")
    assert "1 << 16" in stub and "(h, 1000," in stub
    ns = {}
    exec(stub, ns)
    rd = oracle.Reads(fq)
    table = oracle.Table(21).count(rd.all_seq())
    names, tnf, abd = rd.features(1000, k_tnf=4, k_abd=21, table=table, window=10, vsize=400)
    assert ns["n_rows"] == len(names)
    assert np.array_equal(ns["tnf"].cpu().numpy(), tnf) and np.array_equal(ns["abd"].cpu().numpy(), abd)
