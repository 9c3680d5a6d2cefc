"""`pydca` command line, MSA trimming part (mirror of pydca/main.py:157, :268-281, :397-420,
:464-478): trim_by_refseq and trim_by_gap_size with the reference's flags, output directory and file
names.  The remaining `pydca` sub-commands (contact maps, PDB content, plots) are outside the scope
table (out of scope)."""
import logging
import os
import sys
from argparse import ArgumentParser

from .dca_utilities import dca_utilities
from .msa_trimmer.msa_trimmer import MSATrimmer

logger = logging.getLogger(__name__)
MSA_TRIMMING_SUBCOMMANDS = ['trim_by_refseq', 'trim_by_gap_size']


def execute_from_command_line(msa_file=None, biomolecule=None, the_command=None, refseq_file=None, verbose=False,
                              output_dir=None, max_gap=None, remove_all_gaps=False):
    if verbose:
        logging.basicConfig(level=logging.INFO, format='%(levelname)s %(name)s: %(message)s')
    if the_command.strip() not in MSA_TRIMMING_SUBCOMMANDS:
        raise NotImplementedError('{} is outside the scope table'.format(the_command))
    if the_command.strip() == 'trim_by_refseq':
        msa_trimmer = MSATrimmer(msa_file, biomolecule=biomolecule, refseq_file=refseq_file, max_gap=max_gap)
        columns_to_remove = msa_trimmer.trim_by_refseq(remove_all_gaps=remove_all_gaps)
    else:
        msa_trimmer = MSATrimmer(msa_file, max_gap=max_gap)
        columns_to_remove = msa_trimmer.trim_by_gap_size()
    if not output_dir:
        msa_file_basename, _ext = os.path.splitext(os.path.basename(msa_file))
        output_dir = 'Trimmed_' + msa_file_basename
    dca_utilities.create_directories(output_dir)
    output_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='Trimmed_', postfix='.fa')
    dca_utilities.write_trimmed_msa(output_file_path, msa_trimmer=msa_trimmer, columns_to_remove=columns_to_remove)
    return output_file_path


def run_pydca(argv=None):
    parser = ArgumentParser(prog='pydca')
    subparsers = parser.add_subparsers(dest='subcommand_name')
    p = subparsers.add_parser('trim_by_refseq', help='Removes columns of the MSA that are gaps in the row matching the '
                              'reference sequence (all of them with --remove_all_gaps, else those beyond --max_gap)')
    p.add_argument('--max_gap', type=float)
    p.add_argument('biomolecule')
    p.add_argument('msa_file')
    p.add_argument('refseq_file')
    p.add_argument('--remove_all_gaps', action='store_true')
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--output_dir')
    p = subparsers.add_parser('trim_by_gap_size', help='Removes columns with a gap fraction beyond --max_gap (default 0.5)')
    p.add_argument('--max_gap', type=float)
    p.add_argument('msa_file')
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--output_dir')
    argv = sys.argv[1:] if argv is None else argv
    args = vars(parser.parse_args(args=argv if argv else ['--help']))
    return execute_from_command_line(
        msa_file=args.get('msa_file'), biomolecule=args.get('biomolecule'), the_command=args.get('subcommand_name'),
        refseq_file=args.get('refseq_file'), verbose=args.get('verbose'), output_dir=args.get('output_dir'),
        max_gap=args.get('max_gap'), remove_all_gaps=args.get('remove_all_gaps'))


if __name__ == '__main__':
    run_pydca()
