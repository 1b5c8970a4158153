"""Step-3 constants, same names and values as /root/reference/blaze/config.py:32-34,46,72-75."""
DEFAULT_UMI_SIZE = 12
V2_UMI_SIZE = 10
DEFAULT_BC_SIZE = 16
DEFAULT_GRB_FLANKING_SIZE = 5
DEFAULT_ASSIGNMENT_ED = 2
# Make sure this is smaller than DEFAULT_GRB_FLANKING_SIZE
assert DEFAULT_GRB_FLANKING_SIZE >= DEFAULT_ASSIGNMENT_ED
DEFAULT_ED_FLANKING = DEFAULT_ASSIGNMENT_ED

DEFAULT_GRB_OUT_RAW_BC = 'putative_bc.csv'
DEFAULT_GRB_OUT_WHITELIST = 'whitelist.csv'
DEFAULT_GRB_OUT_FASTQ = "matched_reads.fastq.gz"

# device batching (no counterpart in the reference: its unit of work is --batch-size rows per process)
DEVICE_BATCH_ROWS = 1 << 16       # rows per device submit: small enough that reading, device and writing of successive batches overlap
