"""`mfdca` command line (mirror of pydca/mfdca_main.py:310-394).  every sub-command the reference's
parser exposes (compute_di, compute_fn, compute_params, compute_fi, compute_fij) runs on the GPU."""
import logging
import os
import sys
from argparse import ArgumentParser

from .dca_utilities import dca_utilities
from .meanfield_dca import meanfield_dca
from .sequence_backmapper.sequence_backmapper import SequenceBackmapper

logger = logging.getLogger(__name__)
SUBCOMMANDS = ('compute_di', 'compute_fn', 'compute_params', 'compute_fi', 'compute_fij')


def configure_logging():
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(name)s: %(message)s')


def execute_from_command_line(msa_file=None, biomolecule=None, seqid=None, pseudocount=None, the_command=None,
                              refseq_file=None, verbose=False, output_dir=None, apc=False, ranked_by=None,
                              linear_dist=None, num_site_pairs=None, device=0, devices=None):
    if verbose:
        configure_logging()
    mfdca_instance = meanfield_dca.MeanFieldDCA(msa_file, biomolecule, pseudocount=pseudocount, seqid=seqid, device=device, devices=devices)
    seqbackmapper = None
    if refseq_file:   # do backmapping when a reference sequence file is provided
        seqbackmapper = SequenceBackmapper(alignment_data=mfdca_instance.alignment, refseq_file=refseq_file,
                                           biomolecule=mfdca_instance.biomolecule)
    param_metadata = dca_utilities.mfdca_param_metadata(mfdca_instance)
    if not output_dir:
        msa_file_base_name, _ext = os.path.splitext(os.path.basename(msa_file))
        output_dir = 'MFDCA_output_' + msa_file_base_name
    dca_utilities.create_directories(output_dir)
    if the_command.strip() == 'compute_params':
        fields, couplings = mfdca_instance.compute_params(seqbackmapper=seqbackmapper, ranked_by=ranked_by, linear_dist=linear_dist,
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
    if the_command.strip() == 'compute_fi':
        # pass --pseudocount 0.0 if raw frequencies are desired
        fi = mfdca_instance.get_reg_single_site_freqs()
        metadata = param_metadata + dca_utilities.mfdca_residue_repr_metadata(mfdca_instance.biomolecule)
        fi_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='fi_', postfix='.txt')
        dca_utilities.write_single_site_freqs(fi_file_path, fi, seqs_len=mfdca_instance.sequences_len,
                                              num_site_states=mfdca_instance.num_site_states, metadata=metadata)
        return fi_file_path
    if the_command.strip() == 'compute_fij':
        file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='fij_', postfix='.txt')
        metadata = param_metadata + dca_utilities.mfdca_residue_repr_metadata(mfdca_instance.biomolecule)
        fij = mfdca_instance.get_reg_pair_site_freqs()
        dca_utilities.write_pair_site_freqs(file_path, fij, seqs_len=mfdca_instance.sequences_len,
                                            num_site_states=mfdca_instance.num_site_states, metadata=metadata)
        return file_path
    if the_command.strip() == 'compute_di':
        if apc:
            sorted_DI = mfdca_instance.compute_sorted_DI_APC(seqbackmapper=seqbackmapper)
            score_type = ' MF DI average product corrected (APC)'
            di_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='MFDCA_apc_di_scores_', postfix='.txt')
        else:
            sorted_DI = mfdca_instance.compute_sorted_DI(seqbackmapper=seqbackmapper)
            score_type = 'raw DI'
            di_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='MFDCA_raw_di_scores_', postfix='.txt')
        dca_utilities.write_sorted_dca_scores(di_file_path, sorted_DI, metadata=param_metadata, score_type=score_type)
        return di_file_path
    if apc:
        score_type = 'MFDCA Frobenius norm, average product corrected (APC)'
        sorted_FN = mfdca_instance.compute_sorted_FN_APC(seqbackmapper=seqbackmapper)
        fn_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='MFDCA_apc_fn_scores_', postfix='.txt')
    else:
        score_type = 'MFDCA raw Frobenius norm'
        sorted_FN = mfdca_instance.compute_sorted_FN(seqbackmapper=seqbackmapper)
        fn_file_path = dca_utilities.get_dca_output_file_path(output_dir, msa_file, prefix='MFDCA_raw_fn_scores_', postfix='.txt')
    dca_utilities.write_sorted_dca_scores(fn_file_path, sorted_FN, metadata=param_metadata, score_type=score_type)
    return fn_file_path


def run_meanfield_dca(argv=None):
    parser = ArgumentParser(prog='mfdca')
    subparsers = parser.add_subparsers(dest='subcommand_name')
    for name in SUBCOMMANDS:
        p = subparsers.add_parser(name)
        p.add_argument('biomolecule')
        p.add_argument('msa_file')
        p.add_argument('--seqid', type=float)
        p.add_argument('--pseudocount', type=float)
        p.add_argument('--refseq_file')
        p.add_argument('--output_dir')
        p.add_argument('--verbose', action='store_true')
        p.add_argument('--apc', action='store_true')
        p.add_argument('--device', type=int, default=0, help='GPU index (addition)')
        p.add_argument('--devices', help='comma-separated GPU indices: one rank per GPU for the sequence weights and the pair counts '
                       '(ONE all-reduce of the counts over RCCL); the inverse and the scores run on the first (addition)')
        if name == 'compute_params':
            p.add_argument('--ranked_by', choices=('FN', 'FN_APC', 'DI', 'DI_APC', 'fn', 'fn_apc', 'di', 'di_apc'))
            p.add_argument('--linear_dist', type=int)
            p.add_argument('--num_site_pairs', type=int)
    argv = sys.argv[1:] if argv is None else argv
    args = vars(parser.parse_args(args=argv if argv else ['--help']))
    return execute_from_command_line(
        biomolecule=args.get('biomolecule'), msa_file=args.get('msa_file'), seqid=args.get('seqid'),
        pseudocount=args.get('pseudocount'), the_command=args.get('subcommand_name'), refseq_file=args.get('refseq_file'),
        verbose=args.get('verbose'), output_dir=args.get('output_dir'), apc=args.get('apc'),
        ranked_by=args.get('ranked_by'), linear_dist=args.get('linear_dist'), num_site_pairs=args.get('num_site_pairs'),
        device=args.get('device'), devices=args.get('devices'))


if __name__ == '__main__':
    run_meanfield_dca()
