#!/bin/bash
# timings of the side-by-side path at several shapes
for args in "512 3 2 32" "512 3 2 8" "512 3 2 128" "497 4 1 32" "497 4 1 10" "300 3 2 32" "2048 1 1 32" "200 1 1 256" "1024 1 1 32"; do
  python profiles/batch_run.py $args 4 2>&1 | tail -1
done
