/* tbh_host.h — C ABI of libtbh.so: the host-side halves of the write path that the multi-rank command line
 * (tiebrush_amd/ranks.py, `tiebrush --ranks N`) needs outside the `tiebrush` binary.  Plain pointers and sizes; nothing here
 * touches the GPU.
 *
 * Reference interfaces replaced (under /root/reference/src):
 *   tbh_tag_deflate_part   <-  flushPData's tagging of every output record (tiebrush.cpp:506-525, GSam.h:300-305) followed by
 *                              GSamWriter::write -> sam_write1 (GSam.h:648-653), for one rank's slice of the output
 *   tbh_write_bam_parts    <-  the output header the reference builds while it opens its inputs (TInputFiles::addSam,
 *                              tmerge.cpp:57-147: @CO SAMPLE lines, @PG TieBrush) + GSamWriter's open / close (GSam.h:542-580);
 *                              the record stream is the ranks' parts in rank order (BGZF members concatenate)
 * The reference's only multi-worker scheme ends the same way — one collapsed BAM (tiewrap.py:96-126).
 */
#ifndef TBH_HOST_H_
#define TBH_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBH_ABI_VERSION 1
int tbh_abi_version(void);
const char* tbh_last_error(void); /* message of the calling thread's last failed call */

/* n output records in output order: record g is the raw BAM record (WITHOUT its block_size field) at blob + rec_off[g],
 * rec_len[g] bytes; yc / yx / yd as tbk_groups_out delivers them.  Tags every record as the reference's flushPData does, frames
 * it, deflates the run into whole BGZF members (no BAM header, no EOF member) on `threads` workers (0: the CPU budget) at
 * `level`, and writes the members to `out_path`.  0 on success, -1 otherwise (tbh_last_error). */
int tbh_tag_deflate_part(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, const double* yc, const int64_t* yx,
                         const int32_t* yd, int level, int threads, const char* out_path);

/* The collapsed BAM of a multi-rank run: the header the single-GPU `tiebrush` writes for the same inputs and command line
 * (version, cmd_argc / cmd_argv = the command line as typed, files in input order), then the members of every part file in
 * the order given, then the EOF member.  The parts are deleted when remove_parts != 0.  0 on success, -1 otherwise. */
int tbh_write_bam_parts(const char* out_path, const char* version, int cmd_argc, const char* const* cmd_argv, int n_files, const char* const* files,
                        int n_parts, const char* const* parts, int remove_parts);

/* 1 when the BAM at `path` was written by TieBrush (a @PG line with PN:TieBrush and a VN tag, tmerge.cpp:70-77), 0 when not,
 * -1 when the header cannot be read. */
int tbh_is_tiebrush(const char* path);

#ifdef __cplusplus
}
#endif
#endif /* TBH_HOST_H_ */
