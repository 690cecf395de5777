#!/usr/bin/env python3
"""Drop-in entry point of the autoregressive Transformer-TTS trainer: ``python train.py --hp_file <hparams.py>``
(the reference's command line), running transformer_tts_amd.train."""
from transformer_tts_amd.train import main

if __name__ == "__main__":
    main()
