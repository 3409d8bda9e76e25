"""bench.extra_metrics_sharded on a world of ONE rank over RCCL (the only world a 1-GPU box offers): a smoke run of the N > 1 leg of
bench.py's `extra`.    gpurun -- python tools/sharded_extra.py"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from taxoexpan_amd import TaxoExpan
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(47)
    model = TaxoExpan("PGAT", "WMR", "LBM", **bench.MAG).to(dev)
    print(json.dumps(bench.extra_metrics_sharded(model, dev, 1, 0, n_queries=4096)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
