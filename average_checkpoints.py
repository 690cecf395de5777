#!/usr/bin/env python3
"""Drop-in entry point: ``python average_checkpoints.py --snapshots <dir>/network.epoch* --out <file> --num N`` (the
reference's command line), running transformer_tts_amd.average_checkpoints."""
from transformer_tts_amd.average_checkpoints import main

if __name__ == "__main__":
    main()
