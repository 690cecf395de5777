"""Checkpoint averaging -- the command line and arithmetic of the reference's average_checkpoints.py (:9-45): element-wise
sum of the ``network.epochN`` state_dicts (the last ``--num`` by modification time, or epochs ``--start`` .. ``--end``),
``torch.div`` by the count (integer buffers such as ``num_batches_tracked`` become floating point, as there), ``torch.save``.
The checkpoints written by transformer_tts_amd.train_fastspeech2 / train carry the reference's 217 / AR keys, so the
reference's own tool runs on them unchanged; this is the same host-side tool for an installation without the reference."""
import argparse
import os

import torch


def average(paths, num=None):
    avg = None
    for path in paths:
        print(path)
        states = torch.load(path, map_location=torch.device("cpu"), weights_only=True)
        if avg is None:
            avg = {k: v.clone() for k, v in states.items()}
        else:
            for k in avg.keys():
                avg[k] += states[k]
    num = len(paths) if num is None else num
    for k in avg.keys():
        if avg[k] is not None:
            avg[k] = torch.div(avg[k], num)
    return avg


def select(args):
    if args.start is None and args.end is None:
        print("average {} files from last modified model".format(args.num))
        last = sorted(args.snapshots, key=os.path.getmtime)
        return last[-args.num:]
    if args.start is not None and args.end is not None:
        dirname = os.path.dirname(args.snapshots[0])
        return [os.path.join(dirname, "network.epoch{}".format(epoch)) for epoch in range(args.start, args.end + 1)]
    raise ValueError("give --num, or both --start and --end")


def get_parser():
    parser = argparse.ArgumentParser(description="average models from snapshot")
    parser.add_argument("--snapshots", required=True, type=str, nargs="+")
    parser.add_argument("--out", required=True, type=str)
    parser.add_argument("--num", default=None, type=int)
    parser.add_argument("--start", default=None, type=int)
    parser.add_argument("--end", default=None, type=int)
    parser.add_argument("--backend", default="pytorch", type=str)
    return parser


def main(argv=None):
    args = get_parser().parse_args(argv)
    if args.backend != "pytorch":
        raise ValueError("Incorrect type of backend")
    last = select(args)
    print("average over", last)
    if args.num is None:
        args.num = args.end - args.start + 1
    torch.save(average(last, args.num), args.out)
    print("{} saved.".format(args.out))


if __name__ == "__main__":
    main()
