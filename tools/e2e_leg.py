# bench.py's end-to-end leg by itself (FASTQ file in /dev/shm -> stream files -> output.dna), for looking at its breakdown:
#   python tools/e2e_leg.py [n_reads]        (HARC_AMD_FEED_THREADS / HARC_AMD_FEED_SLICE: reader threads / bytes per pinned slice of the file feeder)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
bench.torch = torch
import harc_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
print(json.dumps(bench.end_to_end_leg(harc_amd, 0, torch.device("cuda", 0), 8, n=n)))
