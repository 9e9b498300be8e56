#!/usr/bin/env python3
"""Entry point with the reference's command line (main.py:13-36: the same 15 flags and defaults)
dispatching to the MI355X trainers.  Extra flags: --synthetic (seeded random patches instead of the image
folders under --data_dir; --steps_per_epoch of them per epoch), --epoch_pretrain and --precision {mixed,bf16x3,fp32}.
Multi-GPU: python -m torch.distributed.run --nproc-per-node N main.py ..."""
import argparse
import os


def _names(text):
    """--train_dataset / --test_dataset: the reference declares them `type=list` (main.py:18-19), which turns a value
    given on the command line into its characters ('DIV2K' -> ['D','I','V','2','K']); here a value is one name or a
    comma-separated list of names."""
    return [n for n in (t.strip() for t in str(text).split(',')) if n]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MI355X-native SR collection (reference-compatible CLI)")
    p.add_argument('--model_name', type=str, default='SRGAN',
                   choices=['SRCNN', 'VDSR', 'ESPCN', 'FSRCNN', 'SRGAN', 'LapSRN', 'EDSR'], help='The type of model')
    p.add_argument('--data_dir', type=str, default='../Data')
    p.add_argument('--train_dataset', type=_names, default=['DIV2K'], help='The name(s) of the training dataset, comma-separated')
    p.add_argument('--test_dataset', type=_names, default=['Set5', 'Set14', 'Urban100'], help='The name(s) of the test dataset, comma-separated')
    p.add_argument('--crop_size', type=int, default=128, help='Size of cropped HR image')
    p.add_argument('--num_threads', type=int, default=4, help='number of threads for data loader to use')
    p.add_argument('--num_channels', type=int, default=3, help='The number of channels to super-resolve')
    p.add_argument('--scale_factor', type=int, default=4, help='Size of scale factor')
    p.add_argument('--num_epochs', type=int, default=100, help='The number of epochs to run')
    p.add_argument('--save_epochs', type=int, default=10, help='Save trained model every this epochs')
    p.add_argument('--batch_size', type=int, default=16, help='training batch size')
    p.add_argument('--test_batch_size', type=int, default=1, help='testing batch size')
    p.add_argument('--save_dir', type=str, default='Result_DIV2K', help='Directory name to save the results')
    p.add_argument('--lr', type=float, default=0.00001)
    p.add_argument('--gpu_mode', type=bool, default=True)
    p.add_argument('--synthetic', action='store_true',
                   help='train / test on seeded random patches instead of the image folders under --data_dir '
                        '(without it a missing folder is an error, as in the reference)')
    p.add_argument('--steps_per_epoch', type=int, default=8, help='--synthetic: batches per epoch')
    p.add_argument('--epoch_pretrain', type=int, default=50, help='SRGAN generator pre-training epochs (srgan.py:179)')
    p.add_argument('--precision', type=str, default='mixed', choices=['mixed', 'bf16x3', 'bf16x6', 'fp32'])
    p.add_argument('--eager', action='store_true', help='launch every kernel of a train step from Python (default: replay the step as a hipGraph)')
    p.add_argument('--sync_bn', action='store_true',
                   help='data-parallel SRGAN: BatchNorm statistics over the GLOBAL batch (all-reduce of the [2C] sums per '
                        'BatchNorm call; default: per-shard statistics), i.e. the single-process step of the reference')
    p.add_argument('--prune_dead_grads', action='store_true',
                   help='SRGAN: skip the two gradient computations of the reference iteration that nothing reads '
                        '(G gradients of the D step, D parameter gradients of the G step); same parameters after every step')
    return check_args(p.parse_args(argv))


def check_args(args):   # main.py:39-57
    args.save_dir = os.path.join(args.save_dir, args.model_name)
    os.makedirs(args.save_dir, exist_ok=True)
    if args.num_epochs < 1:
        print('number of epochs must be larger than or equal to one')
    if args.batch_size < 1:
        print('batch size must be larger than or equal to one')
    return args


def main(argv=None):
    args = parse_args(argv)
    if args is None:
        exit()
    import __graft_entry__
    __graft_entry__.build()
    import pytorch_super_resolution_model_collection_amd as pkg
    from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
    pkg.ops.set_precision(args.precision)
    net = TRAINERS[args.model_name](args)   # main.py:70-89
    net.train()                              # main.py:96
    net.test()                               # main.py:99
    return net


if __name__ == '__main__':
    main()
