"""Golden `result_crc` values of bench.py's DEFAULT workload (BASELINE config 3: 16 x 2048^2, 8 frames; CRC-32 of
every frame's level-0 disparity, see sequence.SequenceRunner.result_crc): what a --gpus N line must print if the
sharded run computed the depth maps of the 1-GPU run. The workload is too large for the CPU oracle; the values
are recorded from a 1-GPU run on the MI355X (synthetic frames rendered on the GPU, so they do not depend on the
host CPU) — either merged from a bench line,

    python tests/golden/gen_bench_crc.py gpurun_out/<bench line>.json

or written by `DERP_RECORD_BASELINE=1 pytest tests/test_gpu_sequence.py -k config3_full` into
gpurun_out/bench_result_crc_cfg2_8.json. The small workloads of the GPU tests are checked against the oracle
itself at test time (tests/test_gpu_bench.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "bench_result_crc.json")


def main():
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    src = json.load(open(sys.argv[1]))
    if "result_crc" in src:  # a bench.py line
        assert src["n_gpus"] == 1
        data["%s_%d" % (src["config"]["rig"], src["config"]["frames"])] = src["result_crc"]
    else:  # the file the full-size test records
        data["cfg2_8"] = src
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps(data, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
