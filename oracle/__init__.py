"""CPU oracle of the aggregation hot path — TEST INFRASTRUCTURE, never imported by the product.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
