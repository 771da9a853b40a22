#!/usr/bin/env python3
"""An independent check of the FETCH_SIZE / WRITE_SIZE corrections of tools/collect_profiles.py: kernels whose
HBM bytes are known exactly -- torch's device-to-device copy and fill of 2 GiB int32 buffers, far larger than L2 + MALL --
run under the same counters.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal_f -- python3 tools/fetch_calibration.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/cal_w -- python3 tools/fetch_calibration.py run
    python3 tools/fetch_calibration.py report /tmp/cal_f /tmp/cal_w  > profiles/<tag>_fetch_calibration.txt
"""
import csv
import glob
import os
import sys

N = 1 << 29                                     # 2 GiB of int32 per buffer


def run():
    import torch
    dev = torch.device("cuda:0")
    a = torch.empty(N, dtype=torch.int32, device=dev).fill_(1)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)                              # reads 2 GiB, writes 2 GiB
    c = torch.empty(N // 2, dtype=torch.int64, device=dev)
    for _ in range(3):
        c.copy_(a.view(torch.int64))            # the same bytes in 8-byte elements
    torch.cuda.synchronize()


def report(fetch_dir, write_dir):
    nbytes = N * 4
    print(f"known: every copy kernel reads {nbytes / 2**30:.0f} GiB and writes {nbytes / 2**30:.0f} GiB; every fill kernel writes {nbytes / 2**30:.0f} GiB")
    for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))[-1]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and float(r["Counter_Value"]) * 1024 > 0.2 * nbytes:
                v = float(r["Counter_Value"]) * 1024
                print(f"{ctr:10s} {v / 2**30:7.3f} GiB = {v / nbytes:5.2f} x the known bytes   {r['Kernel_Name'][:90]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
