"""Data side of the FastSpeech2 training path (mirror of the reference's ``datasets`` package)."""
