"""Parts of bench.py (repo root): the workload and its byte models, the parity gates, the CPU baselines, the rocprofv3 passes, the
GPU-scale sample.  bench.py keeps the command line, the timed region and the JSON line."""
