// Satellites: LDS state (member `sat` of Smem in the satellite builds, NSAT > 0).  See smj_sat.h.
#pragma once
#if NSAT > 0
#define NXV (NVS + 6 * NXS)   // order of the dense Newton system of a step in which satellites are coupled: the main columns + NXS satellites
#ifndef SMJ_SAT_ITEMS
#define SMJ_SAT_ITEMS 20
#endif
#define NIT SMJ_SAT_ITEMS    // row items (a contact's rows / a friction-loss row / a limit row) one satellite can hold in a step
#define NSS 8                 // satellite-satellite contacts of a step
#define NCAND 288             // static-broadphase candidates kept between steps
#define SMJ_SB_SLACK 0.03f    // metres a moving geom may travel before the candidate list is rebuilt
#define NS2 (6 * NSS)         // rows of the second-slot pool
#ifndef NCH
#define NCH 16                // contacts whose cone Hessian is kept per constraint update (contacts in the middle zone of their cone: sliding); beyond it a contact's block is left out of H for that iteration (the step stays exact: gradient and line search are)
#endif
// satellite 6-vectors (one per satellite and field)
enum { SX_V = 0, SX_QA, SX_MA, SX_GRAD, SX_SRCH, SX_MV, SX_G, SX_TMP, SX_N };
enum { ITEM_N = 7, ITEM_SLOT = 8, ITEM_CONTACT = 16, ITEM_MAIN = 32 /* the other body of the contact belongs to the main tree */ };
struct SatMem {
  float x[SX_N][NSAT][6];   // qvel | qacc (qacc_warmstart at the start of the solve) | M qacc | gradient | search | M search | qfrc_smooth | scratch: J'f, right-hand sides
  float q[NSAT][7];
  float Mb[NSAT][21], Hb[NSAT][21];   // mass block and Newton block of the satellite (packed lower triangle, row major); 1-dof satellites: identity padded
  float wax[NSAT][3], wanc[NSAT][3];  // world joint axis / anchor of a hinge or slide satellite
  int jtype[NSAT], ndof[NSAT], body[NSAT];
  int ext[NSAT];            // slot in the dense extension of this step's Newton system (-1: the block is solved on the satellite's own lane)
  int xs[NXS];              // satellites of the extension, slot order
  int nitem[NSAT];
  unsigned short irow[NSAT][NIT];   // first row of the item
  unsigned char iinf[NSAT][NIT], icon[NSAT][NIT];   // rows (ITEM_N), slot, flags | contact index
  int sscon[NSS];           // contacts between two satellites
  float Js[NEFC][6];        // the satellite columns of a constraint row: those of satellite esat[row][0] ...
  float Js2[NS2][6];        // ... and, for the rows of a contact between two satellites, those of esat[row][1]: row e2[row] of this pool
  short e2[NEFC];
  int ns2, nch;             // pool rows handed out this step; cone-Hessian blocks handed out by the last constraint update
  signed char chs[NCON];    // block of the cone-Hessian pool (Smem::u.n.cH) that holds the contact's Hessian, or -1
  signed char esat[NEFC][2];
  // static broadphase candidates (collision_static): (static geom | cache slot << 9) pairs whose boxes came within SMJ_SB_SLACK of
  // each other when the list was last built, and the moving geoms' centres at that time -- the list stays valid until one of them
  // has moved by the slack
  unsigned short cand[NCAND];
  int ncand, cand_ok;
  float refcen[NCG][3];
  short erec[NEFC];         // row record (DevModel::k_rowrec) of a static / limit row -- with satellite rows moved behind the dense rows a row no longer sits at its record's index
};
#endif
