#!/usr/bin/env python3
"""profiles/pmc_vq_lds.json from one rocprofv3 --pmc pass over bench.py (SQ counters of the three codebook-search kernels):
the "LDS hit-rate for the VQ lookup against CDNA4 peak" half of BASELINE.json's reporting sentence.

  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d <dir> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events
  python3 tools/pmc_vq.py <dir> <out.json> [commit] [command ...]

Per kernel (dac_rvq_kernel, rvq_ema_forward_mfma_kernel, rvq_ema_forward_token_kernel, rvq_ema_forward_kernel):
  lds_per_vmem_rd            LDS instructions per global-read instruction: how much of the lookup the LDS serves
  lds_operand_share          LDS / (LDS + global read) instructions = the "hit rate" of the LDS-pinned codebook
  lds_bank_conflict_frac     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles)
  lds_busy_frac_of_peak      SQ_LDS_IDX_ACTIVE / (CUs x kernel cycles): share of the LDS arrays' cycles in use; kernel cycles =
                             GRBM_GUI_ACTIVE / 8 XCDs (the MI355X guide's clock note).  The LDS peak is one access cycle per CU
                             per clock (128 B for ds_read_b32, 256 B for b64 / b128).
Stamped like profiles/pmc_traffic.json (commit + digest of csrc/) so that bench.py can tell whether the kernels changed since."""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic import csrc_digest  # noqa: E402

KEEP = ("dac_rvq_kernel", "rvq_ema_forward_mfma_kernel", "rvq_ema_forward_token_kernel", "rvq_ema_forward_kernel")
CUS = 256


def main(argv):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"{argv[1]}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("mvq::", "")
            name = re.sub(r"\(.*$", "", name)
            if not name.startswith(KEEP):
                continue
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name][r["Counter_Name"]] += 1
    root = Path(__file__).resolve().parent.parent
    out = {"_meta": {"commit": argv[3] if len(argv) > 3 else None, "csrc_sha16": csrc_digest(root),
                     "command": " ".join(argv[4:]) or None,
                     "notes": "one --pmc pass (SQ 6 of 8 slots + GRBM), --kernel-trace only; sums over all launches of the run"}}
    for k, v in sorted(tot.items()):
        lds, rd = v.get("SQ_INSTS_LDS", 0.0), v.get("SQ_INSTS_VMEM_RD", 0.0)
        act, conf = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {c: v[c] for c in sorted(v)}
        e["launches"] = max(cnt[k].values())
        e["lds_per_vmem_rd"] = lds / rd if rd else None
        e["lds_operand_share"] = lds / (lds + rd) if lds + rd else None
        e["lds_bank_conflict_frac"] = conf / act if act else None
        e["lds_busy_frac_of_peak"] = act / (CUS * cyc) if cyc else None
        out[k] = e
        print(f"{k[:48]:48s} x{e['launches']:3d}  LDS/global-read {e['lds_per_vmem_rd'] or 0:7.1f}  LDS share {100 * (e['lds_operand_share'] or 0):5.1f} %  "
              f"bank conflicts {100 * (e['lds_bank_conflict_frac'] or 0):5.2f} %  LDS busy {100 * (e['lds_busy_frac_of_peak'] or 0):5.1f} % of peak")
    json.dump(out, open(argv[2], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv)
