"""Process-wide counter of hipGraph replays.

A replayed graph updates weights and moving statistics in place WITHOUT touching the tensors' Python-side version counters,
so every cache that is keyed by `tensor._version` (the eval-mode WC plan, the convolution weight images, the grouped
coloring tables) also carries this counter: whatever was cached before a replay is rebuilt after it."""
replays = 0
