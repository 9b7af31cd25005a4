#!/bin/bash
bash tools/profile.sh bf16x3 2>&1 | tail -20
bash tools/profile.sh fp32 2>&1 | tail -16
