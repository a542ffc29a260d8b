"""deepsee_amd — MI355X-native (gfx950) implementation of DeepSEE's G+D train-step hot path."""
