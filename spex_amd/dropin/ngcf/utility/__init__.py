"""NGCF's `utility` package (NGCF_SPEX/code/utility/): load_data, batch_test, helper, metrics, Logging."""
