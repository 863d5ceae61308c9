"""CPU oracle for the ezpz LM constraint-solve path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (ezpz_amd/) never does.  See oracle/ezpz_oracle.h for parity status.
"""
