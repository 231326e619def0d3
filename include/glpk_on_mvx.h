/*
 * glpk_on_mvx.h -- the reference-side binding: include this instead of <glpk.h> and link
 * -lmvolps_amd instead of -lglpk (/root/reference/Makefile:2).  Every glp_* name MVOLPS uses
 * (call sites: SURVEY.md section 8(b); bs.h:5, util.h:3, cut.h:4, gmi.cpp:2 include glpk.h)
 * becomes the mvx_* entry point with the same argument list; the GLP_* constants keep GLPK's
 * public values.  tests/boundary/glp_caller.cpp is a caller written against this header in
 * glp_* spelling; tests/test_boundary.py compiles and runs it.
 */
#ifndef GLPK_ON_MVX_H
#define GLPK_ON_MVX_H

#include "mvx.h"
#include "mvx_bnb.h" /* the readers live beside the driver */

typedef mvx_prob glp_prob;
typedef mvx_smcp glp_smcp;

#define GLP_MIN MVX_MIN
#define GLP_MAX MVX_MAX
#define GLP_CV MVX_CV
#define GLP_IV MVX_IV
#define GLP_BV MVX_BV
#define GLP_FR MVX_FR
#define GLP_LO MVX_LO
#define GLP_UP MVX_UP
#define GLP_DB MVX_DB
#define GLP_FX MVX_FX
#define GLP_BS MVX_BS
#define GLP_NL MVX_NL
#define GLP_NU MVX_NU
#define GLP_NF MVX_NF
#define GLP_NS MVX_NS
#define GLP_UNDEF MVX_UNDEF
#define GLP_FEAS MVX_FEAS
#define GLP_INFEAS MVX_INFEAS
#define GLP_NOFEAS MVX_NOFEAS
#define GLP_OPT MVX_OPT
#define GLP_UNBND MVX_UNBND
#define GLP_ON MVX_ON
#define GLP_OFF MVX_OFF
#define GLP_MPS_FILE 2 /* util.cpp:290 */

/* lifecycle */
#define glp_create_prob mvx_create_prob   /* bs.cpp:89,115; util.cpp:33,281 */
#define glp_erase_prob mvx_erase_prob     /* bs.cpp:114 */
#define glp_delete_prob mvx_delete_prob   /* util.cpp:41 */
#define glp_copy_prob mvx_copy_prob       /* bs.cpp:116; util.cpp:34 */
/* build / modify */
#define glp_add_rows mvx_add_rows         /* cut.cpp:23 */
#define glp_add_cols mvx_add_cols         /* (model construction) */
#define glp_set_mat_row mvx_set_mat_row   /* cut.cpp:40 */
#define glp_set_row_bnds mvx_set_row_bnds /* cut.cpp:43 */
#define glp_set_col_bnds mvx_set_col_bnds /* bs.cpp:274,282 */
#define glp_set_obj_coef mvx_set_obj_coef /* util.cpp:55 */
#define glp_set_obj_dir mvx_set_obj_dir   /* util.cpp:58 */
#define glp_set_col_kind mvx_set_col_kind /* (model construction) */
#define glp_set_col_name mvx_set_col_name /* (model construction) */
/* solve */
#define glp_init_smcp mvx_init_smcp
#define glp_simplex mvx_simplex           /* bs.cpp:117,279,287; BranchAndBound.cpp:52,134,141 */
/* query */
#define glp_get_status mvx_get_status     /* util.cpp:423 */
#define glp_get_obj_val mvx_get_obj_val   /* bs.cpp:125,145,156,160,190,210,280,288 */
#define glp_get_col_prim mvx_get_col_prim /* bs.cpp:182,184,232,261; gmi.cpp:37; util.cpp:200,203,436 */
#define glp_get_obj_coef mvx_get_obj_coef /* bs.cpp:190; util.cpp:455 (index 0 = constant term) */
#define glp_get_num_rows mvx_get_num_rows /* gmi.cpp:15 */
#define glp_get_num_cols mvx_get_num_cols /* gmi.cpp:16; bs.cpp:181,250 */
#define glp_get_num_int mvx_get_num_int   /* util.cpp:299 */
#define glp_get_col_kind mvx_get_col_kind /* gmi.cpp:18,51; util.cpp:444 */
#define glp_get_col_stat mvx_get_col_stat /* gmi.cpp:23,50 */
#define glp_get_row_stat mvx_get_row_stat /* gmi.cpp:45 */
#define glp_get_row_ub mvx_get_row_ub     /* gmi.cpp:47 */
#define glp_get_row_lb mvx_get_row_lb     /* util.cpp:378 */
#define glp_get_row_type mvx_get_row_type /* util.cpp:377 */
#define glp_get_col_ub mvx_get_col_ub     /* gmi.cpp:52 */
#define glp_get_col_lb mvx_get_col_lb     /* util.cpp:320 */
#define glp_get_col_type mvx_get_col_type /* util.cpp:319 */
#define glp_get_col_name mvx_get_col_name /* (readers keep column names; glp_copy_prob names flag, util.cpp:34) */
#define glp_get_mat_row mvx_get_mat_row   /* gmi.cpp:84; util.cpp:349 */
#define glp_get_obj_dir mvx_get_obj_dir   /* util.cpp:51 */
#define glp_eval_tab_row mvx_eval_tab_row /* gmi.cpp:36 */
/* I/O + environment */
#define glp_read_lp mvx_read_lp           /* util.cpp:284 */
#define glp_read_mps mvx_read_mps         /* util.cpp:290 */
#define glp_term_out mvx_term_out         /* 2test.cpp:45,53,62; util.cpp:481-482 */
#define glp_version mvx_version           /* util.cpp:278 */

#endif
