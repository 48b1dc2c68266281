"""oracle/ — CPU restatement of the reference's SignNet/BasisNet forward. TEST INFRASTRUCTURE ONLY.

This package is the *checker*: plain-torch fp32 (CPU) functions that restate, op for op,
the forward pass of cptq/SignNet-BasisNet's hot path, each citing the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it.  The product package (`signnet_basisnet_amd/`) never does:
its modules raise if the HIP library is missing.

How it is pinned
----------------
* The reference has NO tests, golden vectors or known-answer files for this path
  (SURVEY.md §4, §8(c)), so there is nothing reference-held to check against:
  **parity unpinned** in that strict sense.
* What it IS checked against (tests/test_oracle_golden.py, fixtures in tests/golden/*.npz):
  outputs of the reference's own module code (`Alchemy/sign_net`, `GINESignNetPyG/core`,
  `GraphPrediction/layers/deepsigns.py`, `LearningFilters/{ign,signbasisnet,models}.py`),
  imported unmodified in the build container by `tests/golden/make_golden.py`.  The
  third-party graph ops those modules call are absent from the image and from
  /root/reference (torch_geometric==2.0.1 GINConv/GINEConv, torch_scatter.scatter,
  torch_sparse.SparseTensor, dgl GINConv — SURVEY.md §8(c)); the generator supplies
  stand-ins that restate their published semantics (tests/golden/ref_shim/).
* Independent cross-check: every neighbourhood aggregation is also compared with a dense
  fp64 adjacency product (tests/test_oracle_props.py) so a stand-in error cannot define truth.

State is passed as a plain `state_dict` (reference key names), so weights are
interchangeable between the reference modules, this oracle and the HIP modules.
"""
