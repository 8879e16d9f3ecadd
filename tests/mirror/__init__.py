"""Python restatements of the reference's HOST logic (clusterer, caller tail, smoother, a BAM reader): checkers of the
C++ host code in svdss_amd/csrc -- test infrastructure, not part of the product package."""
