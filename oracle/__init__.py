"""TEST INFRASTRUCTURE: CPU restatements of the reference hot path (the parity oracle).

Nothing in the product package imports from here.  See lbp_oracle.py / array_oracle.py headers.
"""
