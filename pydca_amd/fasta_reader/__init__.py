from .fasta_reader import (FastaReaderError, alignment_letter2int, get_alignment_from_fasta_file,  # noqa: F401
                           get_alignment_int_form, RES_TO_INT_ALL)
