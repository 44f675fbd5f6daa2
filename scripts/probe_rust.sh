#!/bin/bash
# Reports whether this box can build the Rust side (the intended CPU baseline is the reference's own
# prover; it needs cargo AND the vendored crates, neither of which exists in the authoring image).
for t in cargo rustc; do
  if command -v $t >/dev/null 2>&1; then echo "$t: $($t --version)"; else echo "$t: absent"; fi
done
if command -v cargo >/dev/null 2>&1 && [ -d /root/reference ]; then
  (cd /root/reference && cargo build --offline --release -p bin 2>&1 | tail -3) || true
else
  echo "reference build: not attempted (no cargo or no /root/reference)"
fi
echo "CPU baseline in use: bench.py cpu_baseline kind=port (own C++ restatement, oracle/fastplonk.py), not arkworks"
