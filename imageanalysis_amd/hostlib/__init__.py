"""Host-side helpers the hot-path modules need when they run OUTSIDE the reference tree
(tests, bench, standalone use).  Inside the reference tree the real `props`, `lib.camera`,
`lib.logger` and `lib.smart` are picked up instead (see imageanalysis_amd/_deps.py)."""
