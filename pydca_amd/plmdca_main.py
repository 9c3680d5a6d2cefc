"""`plmdca` command line (mirror of pydca/plmdca_main.py:262-330): same sub-commands, flags,
output directory and file names.  compute_fn, compute_di and compute_params run on the GPU."""
import logging
import os
import sys
from argparse import ArgumentParser

from .dca_utilities import dca_utilities
from .plmdca import plmdca
from .sequence_backmapper.sequence_backmapper import SequenceBackmapper

logger = logging.getLogger(__name__)
DCA_COMPUTATION_SUBCOMMANDS = ('compute_fn', 'compute_di', 'compute_params')


def configure_logging():
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(name)s: %(message)s')


def execute_from_command_line(biomolecule, msa_file, the_command=None, refseq_file=None, seqid=None, lambda_h=None,
                              lambda_J=None, max_iterations=None, apc=False, verbose=False, output_dir=None,
                              num_threads=None, ranked_by=None, linear_dist=None, num_site_pairs=None, device=0,
                              exact_gradient=False, precision=32, devices=None):
    if verbose:
        configure_logging()
    plmdca_instance = plmdca.PlmDCA(msa_file, biomolecule, seqid=seqid, lambda_h=lambda_h, lambda_J=lambda_J,
                                    max_iterations=max_iterations, num_threads=num_threads, verbose=verbose,
                                    device=device, exact_gradient=exact_gradient, precision=precision or 32, devices=devices)
    if the_command in DCA_COMPUTATION_SUBCOMMANDS:
        param_metadata = dca_utilities.plmdca_param_metadata(plmdca_instance)
        if not output_dir:
            msa_file_base_name, _ext = os.path.splitext(os.path.basename(msa_file))
            output_dir = 'PLMDCA_output_' + msa_file_base_name
        dca_utilities.create_directories(output_dir)
        seqbackmapper = None
        if refseq_file:   # do backmapping when a reference sequence file is provided
            seqbackmapper = SequenceBackmapper(msa_file=msa_file, refseq_file=refseq_file,
                                               biomolecule=plmdca_instance.biomolecule)
        if the_command == 'compute_fn':
            if apc:
                score_type = 'PLMDCA Frobenius norm, average product corrected (APC)'
                sorted_FN = plmdca_instance.compute_sorted_FN_APC(seqbackmapper=seqbackmapper)
                fn_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='PLMDCA_apc_fn_scores_', postfix='.txt')
            else:
                score_type = 'PLMDCA Frobenius norm, non-APC (not average product corrected)'
                sorted_FN = plmdca_instance.compute_sorted_FN(seqbackmapper=seqbackmapper)
                fn_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='PLMDCA_raw_fn_scores_', postfix='.txt')
            dca_utilities.write_sorted_dca_scores(fn_file_path, sorted_FN, metadata=param_metadata, score_type=score_type)
            return fn_file_path
        if the_command == 'compute_di':
            if apc:
                score_type = 'PLMDCA  DI scores, average product corrected (APC)'
                sorted_DI = plmdca_instance.compute_sorted_DI_APC(seqbackmapper=seqbackmapper)
                di_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='PLMDCA_apc_di_scores_', postfix='.txt')
            else:
                score_type = 'PLMDCA DI scores, non-APC (not average product corrected)'
                sorted_DI = plmdca_instance.compute_sorted_DI(seqbackmapper=seqbackmapper)
                di_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='PLMDCA_raw_di_scores_', postfix='.txt')
            dca_utilities.write_sorted_dca_scores(di_file_path, sorted_DI, metadata=param_metadata, score_type=score_type)
            return di_file_path
        if the_command == 'compute_params':
            fields, couplings = plmdca_instance.compute_params(seqbackmapper=seqbackmapper, ranked_by=ranked_by, linear_dist=linear_dist,
                                                               num_site_pairs=num_site_pairs)
            fields_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='fields_', postfix='.txt')
            param_metadata.append('#\tTotal number of sites whose fields are extracted: {}'.format(len(fields)))
            dca_utilities.write_fields_csv(fields_file_path, fields, metadata=param_metadata)
            couplings_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='couplings_', postfix='.txt')
            param_metadata.pop()
            param_metadata.append('#\tTotal number of site pairs whose couplings are extracted: {}'.format(len(couplings)))
            if ranked_by is None:
                ranked_by = 'FN_APC'
            param_metadata.append('#\tDCA ranking method used: {}'.format(ranked_by))
            if linear_dist is None:
                linear_dist = 4
            param_metadata.append('#\tMinimum separation beteween site pairs in sequence: |i - j| > {}'.format(linear_dist))
            dca_utilities.write_couplings_csv(couplings_file_path, couplings, metadata=param_metadata)
            return fields_file_path, couplings_file_path
    return None


def run_plm_dca(argv=None):
    parser = ArgumentParser(prog='plmdca')
    subparsers = parser.add_subparsers(dest='subcommand_name')
    for name in DCA_COMPUTATION_SUBCOMMANDS:
        p = subparsers.add_parser(name)
        p.add_argument('biomolecule', help='protein or rna (case insensitive)')
        p.add_argument('msa_file', help='FASTA formatted multiple sequence alignment, one sequence per line')
        p.add_argument('--seqid', type=float)
        p.add_argument('--lambda_h', type=float)
        p.add_argument('--lambda_J', type=float)
        p.add_argument('--max_iterations', type=int)
        p.add_argument('--num_threads', type=int, help='accepted for compatibility; the work runs on the GPU')
        p.add_argument('--refseq_file')
        p.add_argument('--verbose', action='store_true')
        if name != 'compute_params':
            p.add_argument('--apc', action='store_true')
        p.add_argument('--output_dir')
        p.add_argument('--device', type=int, default=0, help='GPU index (addition)')
        p.add_argument('--devices', help='comma-separated GPU indices, e.g. 0,1,2,3,4,5,6,7: one rank per GPU, sequences (or sites) '
                       'sharded, gradients exchanged over RCCL each iteration -- the counterpart of --num_threads (addition)')
        p.add_argument('--exact_gradient', action='store_true', help='exact pseudolikelihood gradient instead of '
                       'the reference semantics (addition)')
        p.add_argument('--precision', type=int, choices=(32, 64), default=32, help='32: float32 as the reference (default); 64: the '
                       'float64 mode, which follows a float64 CPU run of the same optimiser to 1e-4 (addition)')
        if name == 'compute_params':
            p.add_argument('--ranked_by', choices=('FN', 'FN_APC', 'DI', 'DI_APC', 'fn', 'fn_apc', 'di', 'di_apc'))
            p.add_argument('--linear_dist', type=int)
            p.add_argument('--num_site_pairs', type=int)
    argv = sys.argv[1:] if argv is None else argv
    args = vars(parser.parse_args(args=argv if argv else ['--help']))
    return execute_from_command_line(
        args.get('biomolecule'), args.get('msa_file'), the_command=args.get('subcommand_name'),
        refseq_file=args.get('refseq_file'), seqid=args.get('seqid'), lambda_h=args.get('lambda_h'),
        lambda_J=args.get('lambda_J'), max_iterations=args.get('max_iterations'), num_threads=args.get('num_threads'),
        apc=args.get('apc'), output_dir=args.get('output_dir'), verbose=args.get('verbose'),
        ranked_by=args.get('ranked_by'), linear_dist=args.get('linear_dist'), num_site_pairs=args.get('num_site_pairs'),
        device=args.get('device'), exact_gradient=args.get('exact_gradient'), precision=args.get('precision'),
        devices=args.get('devices'))


if __name__ == '__main__':
    run_plm_dca()
