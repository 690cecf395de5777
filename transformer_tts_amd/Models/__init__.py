"""Host-side mirror of the reference's ``Models`` package for the FastSpeech2 training path."""
