#!/usr/bin/env python3
"""Entry point mirroring the reference's `python inference_SPEINet.py [--default_data ...]`."""
from speinet_amd.inference import main

if __name__ == "__main__":
    main()
