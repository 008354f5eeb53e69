"""Model-family adapters written against the INSTALLED transformers (5.x).  The reference's adapters
(framefusion/models/**) target transformers 4.45/4.51 internals and are out of scope; these restate
the same two-call-site protocol for the current API."""
