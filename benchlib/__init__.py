"""bench.py's legs, one module per concern (r6: bench.py had grown into one 500-line main())."""
