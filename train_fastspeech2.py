#!/usr/bin/env python3
"""Drop-in entry point: ``python train_fastspeech2.py --hp_file config/hparams_template.py``
(the reference's command line, README.md:11-12), running transformer_tts_amd.train_fastspeech2."""
from transformer_tts_amd.train_fastspeech2 import main

if __name__ == "__main__":
    main()
