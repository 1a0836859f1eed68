"""Command line front door: `python -m waveforms_amd sample [options] EXPR OUT.npy`.

Same sub-command, option letters and defaults as the reference's CLI
(waveforms/__main__.py:10-31): EXPR is a `wave_eval` expression, sampled on
np.arange(start, stop, 1/sample_rate) -- on the GPU here -- scaled by the amplitude and written
with np.save.  The reference's options only take integers (click infers the type from its
integer defaults); this parser takes any number, integers behave the same.
"""
import argparse
import sys


def _number(text):
    value = float(text)
    return int(value) if value.is_integer() and 'e' not in text.lower() and '.' not in text else value


def _build_parser():
    top = argparse.ArgumentParser(prog='python -m waveforms_amd')
    commands = top.add_subparsers(dest='command', required=True)
    s = commands.add_parser('sample', help='Generate a waveform sample.')
    s.add_argument('-S', '--sample-rate', type=_number, default=44100, help='Sample rate in Hz')
    s.add_argument('-a', '--start', type=_number, default=0, help='Start time in seconds')
    s.add_argument('-l', '--duration', type=_number, default=-1, help='Duration in seconds')
    s.add_argument('-b', '--stop', type=_number, default=1, help='Stop time in seconds')
    s.add_argument('-A', '--amplitude', type=_number, default=1, help='Amplitude')
    s.add_argument('waveform', help='expression understood by wave_eval')
    s.add_argument('output', help='.npy file to write')
    return top


def run_sample(args):
    import numpy as np

    from waveforms_amd import wave_eval

    stop = args.stop
    if args.duration > 0 and stop == 1:        # --duration wins only over the default stop
        stop = args.start + args.duration
    wav = wave_eval(args.waveform)
    wav.start, wav.stop, wav.sample_rate = args.start, stop, args.sample_rate
    np.save(args.output, wav.sample() * args.amplitude)


def main(argv=None):
    args = _build_parser().parse_args(argv)
    if args.command == 'sample':
        run_sample(args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
