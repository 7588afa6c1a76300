// main.cpp -- harc_amd_stage: command-line twin of the reference's stage executables.
//   harc_amd_stage reorder  <basedir> <readlen> [num_thr] [num_chains]     == src/reorder.out  <basedir>   (harc:67)
//   harc_amd_stage encoder  <basedir> <readlen> [num_thr] [num_chains]     == src/encoder.out  <basedir>   (harc:69)
//   harc_amd_stage compress <basedir> <readlen> [num_thr] [num_chains]     == both, stage I -> II handed over in HBM
//   harc_amd_stage pack_order <basedir> <readlen>                          == src/pack_order.out <basedir> (harc:112)
//   harc_amd_stage decoder <basedir> <readlen ignored> <num_thr_e>         == src/decoder.out <basedir> <num_thr> <num_thr_e> (harc:188)
//   harc_amd_stage decoder_preserve <basedir> <ignored> <num_thr_e> [memory_gb]   == unpack_order + decoder_preserve + merge_N (harc:174-185, -m = MAX_BIN_SIZE)
//   harc_amd_stage compressfq <basedir> <readlen> <fastq> [num_thr] [num_chains] [num_steps] [preserve_order] [preserve_quality]
//                                                                          == preprocess + reorder + encoder (+ reorder_quality), FASTQ parsed on the GPU
//   harc_amd_stage preprocess <basedir> <readlen> <fastq>                  == src/preprocess.out <fastq> <basedir> .. <readlen> (harc:50)
//   harc_amd_stage compressfq_shard <basedir> <readlen> <fastq> <num_thr> <num_chains> <num_steps> <preserve_order> <preserve_quality>
//                                   <world> <rank> <comm_spec> [device] [replicate]    one rank of `./harc -c -g <world>` (one process per GPU)
//   harc_amd_stage merge_shards <basedir> <world>                          the whole-job files of the archive from the rank parts
// readlen / num_thr arrive as arguments instead of the compile-time macros of src/config.h (harc:52-63).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/harc_amd.h"

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s reorder|encoder|compress|pack_order <basedir> <readlen> [num_thr] [num_chains]\n", argv[0]); return 2; }
    // RCCL between processes needs dmabuf IPC on this driver stack (hipIpcGetMemHandle fails otherwise): set before the first HIP call
    // of the process, which is inside the library.  An explicit setting of the caller wins.
    if (!strcmp(argv[1], "compressfq_shard")) setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    if (!strcmp(argv[1], "merge_shards")) {
        const int rcm = harc_amd_merge_shard_files(argv[2], atoi(argv[3]));
        if (rcm != 0) { fprintf(stderr, "harc_amd_stage merge_shards failed (%d): %s\n", rcm, harc_amd_last_error()); return 1; }
        return 0;
    }
    harc_amd_params P;
    int rl = atoi(argv[3]);
    if (!strncmp(argv[1], "decoder", 7) && (rl < 1 || rl > 255)) rl = 100;     // the decoder takes readlen from read_meta.txt (decoder.cpp:324-333)
    if (harc_amd_default_params(rl, &P) != 0) { fprintf(stderr, "%s\n", harc_amd_last_error()); return 1; }
    if (strcmp(argv[1], "preprocess") && strncmp(argv[1], "decoder", 7) && strncmp(argv[1], "compressfq", 10)) {
        if (argc > 4) P.num_thr = atoi(argv[4]);
        if (argc > 5) P.num_chains = atoi(argv[5]);
        if (argc > 6) P.num_steps = atoi(argv[6]);
    }
    int rc;
    if (!strcmp(argv[1], "preprocess")) { if (argc < 5) { fprintf(stderr, "preprocess needs <fastq>\n"); return 2; } rc = harc_amd_preprocess_files(argv[4], argv[2], atoi(argv[3])); }
    else if (!strcmp(argv[1], "compressfq")) {                    // compressfq <basedir> <readlen> <fastq> [num_thr] [num_chains] [num_steps]
        if (argc < 5) { fprintf(stderr, "compressfq needs <fastq>\n"); return 2; }
        if (argc > 5) P.num_thr = atoi(argv[5]);
        if (argc > 6) P.num_chains = atoi(argv[6]);
        if (argc > 7) P.num_steps = atoi(argv[7]);
        const int po = argc > 8 && !strcmp(argv[8], "True"), pq = argc > 9 && !strcmp(argv[9], "True");     // preprocess.out's argv[3], argv[4] (harc:50)
        rc = harc_amd_compress_fastq_files_ex(&P, argv[4], argv[2], po, pq);
    }
    else if (!strcmp(argv[1], "compressfq_shard")) {
        if (argc < 13) { fprintf(stderr, "compressfq_shard needs <fastq> <num_thr> <num_chains> <num_steps> <p> <q> <world> <rank> <comm_spec> [device]\n"); return 2; }
        P.num_thr = atoi(argv[5]); P.num_chains = atoi(argv[6]); P.num_steps = atoi(argv[7]);
        const int po = !strcmp(argv[8], "True"), pq = !strcmp(argv[9], "True"), world = atoi(argv[10]), rank = atoi(argv[11]);
        P.device = argc > 13 ? atoi(argv[13]) : rank;                 // one process per GPU: rank r drives device r unless told otherwise
        P.reads_per_chain = 1024;                                     // a bucket shard is fragmented already (multi-GPU)
        // [mode] after [device]: "replicate" = design (R) (reads all-gathered, chains partitioned, single-GPU bytes); default: minimizer-bucket shards
        // anything else is refused: a typo must not silently give the archive with the larger consensus
        if (argc > 14 && strcmp(argv[14], "replicate") && strcmp(argv[14], "bucket")) { fprintf(stderr, "compressfq_shard: mode '%s' is neither 'bucket' nor 'replicate'\n", argv[14]); return 2; }
        const bool repl = argc > 14 && !strcmp(argv[14], "replicate");
        if (repl) { P.reads_per_chain = 0; rc = harc_amd_compress_fastq_replicated_files(&P, argv[4], argv[2], po, pq, world, rank, argv[12]); }
        else rc = harc_amd_compress_fastq_shard_files(&P, argv[4], argv[2], po, pq, world, rank, argv[12]);
    }
    else if (!strcmp(argv[1], "decoder_preserve")) { if (argc > 5) P.decode_memory_gb = atoi(argv[5]); rc = harc_amd_decoder_preserve_files(&P, argv[2], argc > 4 ? atoi(argv[4]) : 1); }   // [memory]: -m of harc:174
    else if (!strcmp(argv[1], "decoder")) rc = harc_amd_decoder_files(&P, argv[2], argc > 4 ? atoi(argv[4]) : 1);
    else if (!strcmp(argv[1], "reorder")) rc = harc_amd_reorder_files(&P, argv[2]);
    else if (!strcmp(argv[1], "encoder")) rc = harc_amd_encoder_files(&P, argv[2]);
    else if (!strcmp(argv[1], "compress")) rc = harc_amd_compress_files(&P, argv[2]);
    else if (!strcmp(argv[1], "pack_order")) rc = harc_amd_pack_order_files(&P, argv[2]);
    else { fprintf(stderr, "unknown stage %s\n", argv[1]); return 2; }
    if (rc != 0) { fprintf(stderr, "harc_amd_stage %s failed (%d): %s\n", argv[1], rc, harc_amd_last_error()); return 1; }   // non-zero aborts `set -e` (harc:2)
    return 0;
}
