#!/usr/bin/env python3
"""Drop-in entry point: ``python test_fastspeech2.py --load_name <ckpt> --test_script <script>`` (the reference's
synthesis command line), running transformer_tts_amd.test_fastspeech2."""
from transformer_tts_amd.test_fastspeech2 import main

if __name__ == "__main__":
    main()
